!> TEST DRIVER: what the unchanged host does, with the reference's names only -- no `_gpu` name in sight.
!    build_Hv_sector (ED_HAMILTONIAN.f90:43-168):  Hsector/Hstatus, call ed_buildh_main(isector), spHtimesV_p => spMatVec_main
!    ed_diag_d (ED_DIAG.f90:143-209):              call sp_lanc_eigh / sp_eigh(spHtimesV_p, ...)  [SF_SP_LINALG stand-in]
!    delete_Hv_sector (174-222):                   call sp_delete_matrix(spH0d) ...
!  Config 1 (Ns=6, sector (3,3)) in stored and on-the-fly mode, the dense Hmat form, and the 2-orbital replica bath with
!  Jx/Jp of the golden record replica_Nbath2_J.
program host_like_driver
  USE ED_HAMILTONIAN_COMMON
  USE ED_HAMILTONIAN_SPARSE_HxV
  USE ED_HAMILTONIAN_DIRECT_HxV
  USE SF_SP_LINALG
  use EDGPU_LIVE, only: live_status
  implicit none
  real(8), allocatable :: v(:), Hv(:), vect(:), eval(:), evec(:,:), Hmat(:,:), alanc(:), blanc(:)
  real(8) :: egs
  integer :: isector, N, i, imode

  ! ---- config 1: 1 orbital, Nbath=5, normal bath, default initial bath -------------------------------------------
  Norb = 1; Nspin = 1; Nbath = 5; bath_type = "normal"; ed_total_ud = .true.
  call setup_dimensions()
  allocate(impHloc(Nspin,Nspin,Norb,Norb)); impHloc = 0d0
  allocate(dmft_bath%e(Nspin,Norb,Nbath), dmft_bath%v(Nspin,Norb,Nbath))
  dmft_bath%e(1,1,:) = [-2d0, -1d0, 0d0, 1d0, 2d0]
  dmft_bath%v = 1d0/sqrt(dble(Nbath))
  isector = 1 + 3*(Ns_Orb+1) + 3
  do imode = 1, 2
     ed_sparse_H = (imode == 1)
     call host_build(isector)
     N = getDim(isector)
     allocate(v(N), Hv(N), vect(N))
     do i = 1, N
        v(i) = sin(dble(i))
     end do
     call spHtimesV_p(N, v, Hv)
     write(*,"(A,L1,A,I6)") "ed_sparse_H=", ed_sparse_H, " N=", N
     write(*,"(A,ES24.16)") "sum_Hv=", sum(Hv)
     write(*,"(A,ES24.16)") "v_Hv=", dot_product(v, Hv)
     write(*,"(A,3ES24.16)") "Hv_1_3=", Hv(1:3)
     vect = 0d0
     call sp_lanc_eigh(spHtimesV_p, egs, vect, min(N,512), iverbose=.false., threshold=1d-14)
     write(*,"(A,ES24.16)") "E0=", egs
     allocate(eval(2), evec(N,2)); evec = 0d0
     call sp_eigh(spHtimesV_p, eval, evec, 20, 300, tol=1d-12, iverbose=.false.)
     write(*,"(A,2ES24.16)") "E0E1=", eval
     call sp_dvdson_eigh(spHtimesV_p, eval, evec, nitermax=300, tol=1d-11)
     write(*,"(A,2ES24.16)") "E0E1_dvdson=", eval
     allocate(alanc(20), blanc(20))
     call sp_lanc_tridiag(spHtimesV_p, v/sqrt(dot_product(v,v)), alanc, blanc)
     write(*,"(A,2ES24.16)") "alanc_1_2=", alanc(1:2)
     call host_delete()
     deallocate(v, Hv, vect, eval, evec, alanc, blanc)
  end do
  ! dense form: build_Hv_sector(isector,Hmat) (ED_HAMILTONIAN.f90:128-137)
  N = getDim(isector); allocate(Hmat(N,N))
  call host_build(isector, Hmat)
  write(*,"(A,3ES24.16)") "Hmat_trace_sum_11=", sum([(Hmat(i,i), i=1,N)]), sum(Hmat), Hmat(1,1)
  write(*,"(A,L1)") "Hmat_symmetric=", maxval(abs(Hmat - transpose(Hmat))) == 0d0
  call host_delete(); deallocate(Hmat)

  ! ---- a rank the shrunk communicator leaves out: sector (3,0) has DimDw = 1 < MpiSize = 4, build_Hv_sector gives ranks 1..3
  !      MpiComm = MPI_COMM_NULL and leaves their MpiRank/MpiSize alone (ED_HAMILTONIAN.f90:65-93); they still call the
  !      builder and the solver (ED_DIAG.f90:143-185), which must return at once without creating a GPU sector
  MpiStatus = .true.; MpiComm = MPI_COMM_NULL; MpiRank = 2; MpiSize = 4
  isector = 1 + 3*(Ns_Orb+1) + 0
  do imode = 1, 2
     ed_sparse_H = (imode == 1)
     call host_build(isector)
     allocate(eval(1), evec(0,1), vect(0), alanc(3), blanc(3))        ! vecDim_Hv_sector = 0 on such a rank
     eval = -77d0; alanc = -77d0
     call sp_eigh(MpiComm, spHtimesV_p, eval, evec, 1, 10, tol=1d-12)
     call sp_lanc_eigh(MpiComm, spHtimesV_p, eval(1), vect, 10)
     call sp_lanc_tridiag(MpiComm, spHtimesV_p, vect, alanc, blanc)
     write(*,"(A,L1,A,L1)") "left_out_rank_untouched=", (.not. live_status) .and. eval(1) == -77d0 .and. alanc(1) == -77d0, " sparse=", ed_sparse_H
     call host_delete()
     deallocate(eval, evec, vect, alanc, blanc)
  end do
  N = getDim(isector); allocate(Hmat(N,N)); Hmat = -77d0
  call host_build(isector, Hmat)                                      ! LAPACK form: ed_buildh_main(isector,Hmat) returns too (35-37)
  write(*,"(A,L1)") "left_out_rank_hmat_untouched=", (.not. live_status) .and. all(Hmat == -77d0)
  call host_delete(); deallocate(Hmat)
  MpiStatus = .false.; MpiRank = 0; MpiSize = 1

  ! ---- 2-orbital replica bath, H_basis={sigma_z}, lambda_imp=1, Jx=Jp=0.25 (golden record replica_Nbath2_J) ------
  deallocate(impHloc, dmft_bath%e, dmft_bath%v)
  Norb = 2; Nbath = 2; bath_type = "replica"; Uloc = [2d0, 2d0, 0d0, 0d0, 0d0]; Ust = 1.5d0; Jh = 0.25d0; Jx = 0.25d0; Jp = 0.25d0
  call setup_dimensions()
  allocate(H_basis(1)); allocate(H_basis(1)%O(Nspin,Nspin,Norb,Norb)); H_basis(1)%O = 0d0
  H_basis(1)%O(1,1,1,1) = 1d0; H_basis(1)%O(1,1,2,2) = -1d0
  allocate(impHloc(Nspin,Nspin,Norb,Norb)); impHloc = H_basis(1)%O              ! lambda_imp = 1
  allocate(dmft_bath%item(Nbath))
  do i = 1, Nbath
     allocate(dmft_bath%item(i)%v(Nspin), dmft_bath%item(i)%lambda(1))
     dmft_bath%item(i)%v = 1d0/sqrt(dble(Nbath))
     dmft_bath%item(i)%lambda(1) = dble(i) * 2d0 / dble(Nbath)               ! default replica init: rescale(k) * lambda_imp
  end do
  ed_sparse_H = .true.
  isector = 1 + 3*(Ns_Orb+1) + 3
  call host_build(isector)
  N = getDim(isector); allocate(v(N), Hv(N))
  do i = 1, N
     v(i) = sin(dble(i))
  end do
  call spHtimesV_p(N, v, Hv)
  write(*,"(A,I6)") "replica N=", N
  write(*,"(A,ES24.16)") "replica_sum_Hv=", sum(Hv)
  write(*,"(A,ES24.16)") "replica_v_Hv=", dot_product(v, Hv)
  call host_delete()

contains

  !> the steps of build_Hv_sector the drop-ins rely on, restated: sector globals, then the builder call and the pointer
  !  binding written as in ED_HAMILTONIAN.f90:128-166
  subroutine host_build(isector, Hmat)
    integer                         :: isector
    real(8),dimension(:,:),optional :: Hmat
    integer :: Nups(Ns_Ud), Ndws(Ns_Ud), k
    Hsector = isector
    Hstatus = .true.
    allocate(Hs(2*Ns_Ud), DimUps(Ns_Ud), DimDws(Ns_Ud))
    call get_Nup(isector, Nups); call get_Ndw(isector, Ndws)
    do k = 1, Ns_Ud
       DimUps(k) = binom(Ns_Orb, Nups(k)); DimDws(k) = binom(Ns_Orb, Ndws(k))
    end do
    DimUp = product(DimUps); DimDw = product(DimDws); Dim = getDim(isector)
    if (present(Hmat)) then
       spHtimesV_p => null()
       call ed_buildh_main(isector, Hmat)
       return
    end if
    select case (ed_sparse_H)
    case (.true.)
       spHtimesV_p => spMatVec_main
       call ed_buildh_main(isector)
    case (.false.)
       spHtimesV_p => directMatVec_main
    end select
  end subroutine host_build

  subroutine host_delete()
    integer :: iud
    deallocate(Hs, DimUps, DimDws)
    Hsector = 0
    Hstatus = .false.
    call sp_delete_matrix(spH0d)
    if (DimPh > 1) call sp_delete_matrix(spH0e_eph)
    if (Jhflag) call sp_delete_matrix(spH0nd)
    do iud = 1, Ns_Ud
       call sp_delete_matrix(spH0ups(iud))
       call sp_delete_matrix(spH0dws(iud))
    end do
    spHtimesV_p => null()
    iter = 0
  end subroutine host_delete

end program host_like_driver
