!> BASELINE config 1 from the Fortran side: 1 orbital, Nbath=5 (Ns=6), sector (3,3), default
!  initial bath (ED_BATH/dmft_aux.f90:110-133).  Binds the GPU operator to a procedure pointer
!  with the reference's abstract interface and prints the SURVEY 8c known-answer quantities.
program demo_c1
  use, intrinsic :: iso_c_binding
  use EDGPU_C_API
  use ED_HAMILTONIAN_GPU
  use ED_SPARSE_MATRIX
  implicit none
  abstract interface
     subroutine dd_sparse_HxV(Nloc, v, Hv)      ! ED_VARS_GLOBAL.f90:75-81
       integer                 :: Nloc
       real(8), dimension(Nloc) :: v
       real(8), dimension(Nloc) :: Hv
     end subroutine dd_sparse_HxV
  end interface
  procedure(dd_sparse_HxV), pointer :: spHtimesV_p => null()
  integer, parameter :: Norb = 1, Nspin = 1, Nbath = 5
  real(8) :: impHloc(Nspin,Nspin,Norb,Norb), diag_hybr(Nspin,Norb,Nbath), bath_diag(Nspin,Norb,Nbath)
  real(8) :: Uloc(5), g_ph(5), egs, eval2(2)
  real(8), allocatable :: evec2(:,:)
  real(8), allocatable :: v(:), Hv(:), vect(:), alanc(:), blanc(:)
  integer :: N, i, nlanc, isparse
  logical :: sparse_H
  type(sparse_matrix_csr) :: spA
  type(sector_gpu) :: Ssrc
  real(8) :: norm2, hist(4), agf(20), bgf(20), achi(20), bchi(20)
  real(8), allocatable :: gs(:), tw(:)
  integer :: nused
  real(8) :: dense(3,3), xv(3), yv(3)
  ! device-resident flow through the C ABI (vectors as type(c_ptr) device addresses)
  type(sector_gpu) :: Sdst
  type(c_ptr) :: gs_dev
  real(8), pointer, contiguous :: gs_f(:)
  real(c_double) :: e_dev, n2_dev, agd(20), bgd(20)
  integer(c_int32_t) :: nl_dev, nu_dev
  integer(c_int) :: rc
  impHloc = 0d0
  bath_diag(1,1,:) = [-2d0, -1d0, 0d0, 1d0, 2d0]
  diag_hybr = 1d0/sqrt(dble(Nbath))
  Uloc = 0d0; Uloc(1) = 2d0; g_ph = 0d0
  do isparse = 1, 2
     sparse_H = (isparse == 1)
     call build_Hv_sector_gpu(Norb, Nspin, Nbath, 0, "normal", .true., sparse_H, .true., [3], [3], &
          Uloc, 0d0, 0d0, 0d0, 0d0, 0d0, 0d0, g_ph, impHloc, diag_hybr, bath_diag)
     spHtimesV_p => spMatVec_gpu
     N = vecDim_Hv_sector_gpu()
     if (.not. allocated(v)) allocate(v(N), Hv(N), vect(N), alanc(20), blanc(20))
     do i = 1, N
        v(i) = sin(dble(i))
     end do
     call spHtimesV_p(N, v, Hv)
     write(*,"(A,L1,A,I6)") "ed_sparse_H=", sparse_H, " N=", N
     write(*,"(A,ES24.16)") "sum_Hv=", sum(Hv)
     write(*,"(A,ES24.16)") "v_Hv=", dot_product(v, Hv)
     write(*,"(A,3ES24.16)") "Hv_1_3=", Hv(1:3)
     vect = 0d0
     call sp_lanc_eigh_gpu(egs, vect, min(N, 512), threshold=1d-14, nlanc=nlanc)
     write(*,"(A,F20.14,A,I4)") "E0=", egs, " nlanc=", nlanc
     call sp_lanc_tridiag_gpu(v, alanc, blanc)
     write(*,"(A,2ES24.16)") "alanc_1_2=", alanc(1:2)
     if (.not. allocated(evec2)) allocate(evec2(N,2))
     call sp_eigh_gpu(eval2, evec2, 20, 300, tol=1d-12)
     write(*,"(A,2F20.14)") "E0E1=", eval2
     call sp_dvdson_eigh_gpu(eval2, evec2, 300, tol=1d-11)
     write(*,"(A,2F20.14)") "E0E1_dvdson=", eval2
     spHtimesV_p => null()
     call delete_Hv_sector_gpu()
  end do
  ! callers either side of the product: ground state of (3,3), its impurity weights, one particle (c^+_up -> sector
  ! (4,3)) Green's-function leg, a spin-susceptibility leg and the twin hand-off, all through the Fortran wrappers
  call build_Hv_sector_gpu(Norb, Nspin, Nbath, 0, "normal", .true., .true., .true., [3], [3], &
       Uloc, 0d0, 0d0, 0d0, 0d0, 0d0, 0d0, g_ph, impHloc, diag_hybr, bath_diag)
  allocate(gs(N), tw(N)); gs = 0d0
  call sp_lanc_eigh_gpu(egs, gs, min(N, 512), threshold=1d-14)
  call imp_histogram_gpu(gs, hist)
  write(*,"(A,ES24.16)") "docc=", hist(4)
  write(*,"(A,ES24.16)") "hist_sum=", sum(hist)
  call lanc_chi_leg_gpu([0.5d0], [-0.5d0], gs, achi, bchi, norm2, nused)
  write(*,"(A,ES24.16)") "chi_norm2=", norm2
  write(*,"(A,2ES24.16)") "chi_alanc_1_2=", achi(1:2)
  call twin_reorder_gpu(gs, tw)
  write(*,"(A,ES24.16)") "twin_overlap=", abs(dot_product(gs, tw))
  call delete_Hv_sector_gpu()
  call create_sector_gpu(Ssrc, Norb, Nspin, Nbath, 0, "normal", .true., .false., .true., [3], [3], &
       Uloc, 0d0, 0d0, 0d0, 0d0, 0d0, 0d0, g_ph, impHloc, diag_hybr, bath_diag)
  call build_Hv_sector_gpu(Norb, Nspin, Nbath, 0, "normal", .true., .true., .true., [4], [3], &
       Uloc, 0d0, 0d0, 0d0, 0d0, 0d0, 0d0, g_ph, impHloc, diag_hybr, bath_diag)
  call lanc_gf_leg_gpu(Ssrc, 1, 1, .true., gs, agf, bgf, norm2, nused)
  write(*,"(A,ES24.16)") "gf_norm2=", norm2
  write(*,"(A,2ES24.16)") "gf_alanc_1_2=", agf(1:2)
  write(*,"(A,ES24.16)") "gf_blanc_2=", bgf(2)
  call delete_Hv_sector_gpu()
  ! the same leg with the state resident in HBM: zero start vector (the library generates its own, ED_DIAG.f90:147), ground
  ! state searched in place, handed to the leg as a device address -- nothing crosses PCIe but the coefficients
  call create_sector_gpu(Sdst, Norb, Nspin, Nbath, 0, "normal", .true., .true., .true., [4], [3], &
       Uloc, 0d0, 0d0, 0d0, 0d0, 0d0, 0d0, g_ph, impHloc, diag_hybr, bath_diag)
  rc = edgpu_malloc(gs_dev, int(N, c_size_t) * 8_c_size_t)
  if (rc == 0) rc = edgpu_memset(gs_dev, 0_c_int32_t, int(N, c_size_t) * 8_c_size_t)
  if (rc == 0) rc = edgpu_lanczos_gs_device(Ssrc%h, 400_c_int32_t, 1d-14, 10_c_int32_t, e_dev, gs_dev, nl_dev)
  write(*,"(A,F20.14,A,I3)") "E0_device=", e_dev, " rc=", rc
  call c_f_pointer(gs_dev, gs_f, [N])          ! a Fortran name for the device address; never dereferenced on the host
  rc = edgpu_gf_lanczos(Ssrc%h, Sdst%h, 1_c_int32_t, 1_c_int32_t, 1_c_int32_t, gs_f, 20_c_int32_t, 1d-12, n2_dev, agd, bgd, nu_dev)
  write(*,"(A,ES24.16,A,I3)") "gf_norm2_device=", n2_dev, " rc=", rc
  write(*,"(A,2ES24.16)") "gf_alanc_1_2_device=", agd(1:2)
  rc = edgpu_free(gs_dev)
  call delete_sector_gpu(Sdst)
  call delete_sector_gpu(Ssrc)
  ! sparse container: duplicate-summing insert, dense dump, product on the device
  call sp_init_matrix(spA, 3)
  call sp_insert_element(spA, 2d0, 1, 1); call sp_insert_element(spA, -1d0, 1, 3)
  call sp_insert_element(spA, 0.5d0, 1, 3); call sp_insert_element(spA, 4d0, 3, 2)
  dense = 0d0; call sp_dump_matrix(spA, dense)
  xv = [1d0, 2d0, 3d0]; yv = 0d0; call sp_matvec(spA, xv, yv)
  write(*,"(A,3F10.5)") "sp_dense_row1=", dense(1,:)
  write(*,"(A,3F10.5)") "sp_matvec=", yv
  call sp_delete_matrix(spA)
end program demo_c1
