!> Drop-in operator module for the Fortran host: same procedure roles as ED_HAMILTONIAN
!  (build_Hv_sector / delete_Hv_sector / vecDim_Hv_sector, ED_HAMILTONIAN.f90:10-13) and a
!  matvec with the dd_sparse_HxV signature (ED_VARS_GLOBAL.f90:75-81) so that
!     spHtimesV_p => spMatVec_gpu
!  works unchanged, plus the two SciFortran entry points ED_DIAG / ED_GF_NORMAL call with it
!  (sp_lanc_eigh ED_DIAG.f90:177-185, sp_lanc_tridiag ED_GF_NORMAL.f90:232-237) running the
!  whole loop in HBM.  One sector is live at a time, as in the reference.
!
!  The reference's own modules cannot be compiled in this repository (SciFortran is absent),
!  so the inputs the reference holds in ED_INPUT_VARS / ED_VARS_GLOBAL are passed explicitly;
!  INTEGRATION.md shows the three-line call sites inside ED_HAMILTONIAN.f90 / ED_DIAG.f90.
module ED_HAMILTONIAN_GPU
  use, intrinsic :: iso_c_binding
  use EDGPU_C_API
  implicit none
  private

  type(c_ptr), save      :: Hhandle = c_null_ptr
  logical, save          :: Hstatus = .false.
  type(edgpu_info), save :: Hinfo

  public :: build_Hv_sector_gpu
  public :: delete_Hv_sector_gpu
  public :: vecDim_Hv_sector_gpu
  public :: getDim_Hv_sector_gpu
  public :: spMatVec_gpu
  public :: sp_lanc_eigh_gpu
  public :: sp_lanc_tridiag_gpu
  public :: sp_eigh_gpu
  public :: get_sector_map_gpu

contains

  subroutine edgpu_check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    if (rc /= 0) then
       write(*,"(A)") trim(where)//": "//edgpu_error_message()
       stop "edgpu error"                       ! the reference aborts with `stop "<msg>"`
    end if
  end subroutine edgpu_check

  !> build_Hv_sector(isector): ED_HAMILTONIAN.f90:43-168.  Arrays are the reference's own
  !  (column-major, same index order); Nups/Ndws are get_Nup/get_Ndw of the sector.
  subroutine build_Hv_sector_gpu(Norb, Nspin, Nbath, Nph, bath_type, ed_total_ud, ed_sparse_H, hfmode, &
       Nups, Ndws, Uloc, Ust, Jh, Jx, Jp, xmu, w0_ph, g_ph, impHloc, diag_hybr, bath_diag, hbath_tmp, &
       MpiRank, MpiSize, explicit_csr)
    integer, intent(in)                   :: Norb, Nspin, Nbath, Nph
    character(len=*), intent(in)          :: bath_type
    logical, intent(in)                   :: ed_total_ud, ed_sparse_H, hfmode
    integer, intent(in)                   :: Nups(:), Ndws(:)
    real(8), intent(in)                   :: Uloc(:), Ust, Jh, Jx, Jp, xmu, w0_ph, g_ph(:)
    real(8), intent(in), target, contiguous :: impHloc(:,:,:,:)        ! (Nspin,Nspin,Norb,Norb)
    real(8), intent(in), target, contiguous :: diag_hybr(:,:,:)        ! (Nspin,Norb,Nbath)
    real(8), intent(in), target, contiguous :: bath_diag(:,:,:)        ! (Nspin,Nfoo,Nbath)
    real(8), intent(in), target, contiguous, optional :: hbath_tmp(:,:,:,:,:) ! replica
    integer, intent(in), optional         :: MpiRank, MpiSize
    logical, intent(in), optional         :: explicit_csr
    type(edgpu_desc) :: d
    integer :: n
    if (Hstatus) stop "build_Hv_sector_gpu ERROR: a sector is already built (call delete_Hv_sector_gpu)"
    d%abi_version = EDGPU_ABI_VERSION
    d%Norb = Norb; d%Nspin = Nspin; d%Nbath = Nbath; d%Nph = Nph
    select case (trim(bath_type))
    case ("hybrid");  d%bath_type = EDGPU_BATH_HYBRID
    case ("replica"); d%bath_type = EDGPU_BATH_REPLICA
    case default;     d%bath_type = EDGPU_BATH_NORMAL
    end select
    d%ed_total_ud = merge(1, 0, ed_total_ud)
    d%storage = merge(EDGPU_STORE_KRON, EDGPU_STORE_DIRECT, ed_sparse_H)
    if (present(explicit_csr)) then
       if (explicit_csr .and. ed_sparse_H) d%storage = EDGPU_STORE_CSR
    end if
    d%hfmode = merge(1, 0, hfmode)
    d%Nfoo = size(bath_diag, 2)
    d%nups = 0; d%ndws = 0
    n = size(Nups); d%nups(1:n) = Nups; d%ndws(1:n) = Ndws
    d%mpi_rank = 0; d%mpi_size = 1
    if (present(MpiRank)) d%mpi_rank = MpiRank
    if (present(MpiSize)) d%mpi_size = MpiSize
    d%device = -1; d%reserved = 0
    d%Uloc = 0d0; d%Uloc(1:min(size(Uloc),5)) = Uloc(1:min(size(Uloc),5))
    d%Ust = Ust; d%Jh = Jh; d%Jx = Jx; d%Jp = Jp; d%xmu = xmu; d%w0_ph = w0_ph
    d%g_ph = 0d0; d%g_ph(1:min(size(g_ph),5)) = g_ph(1:min(size(g_ph),5))
    d%impHloc = c_loc(impHloc); d%diag_hybr = c_loc(diag_hybr); d%bath_diag = c_loc(bath_diag)
    d%hbath = c_null_ptr
    if (present(hbath_tmp)) d%hbath = c_loc(hbath_tmp)
    call edgpu_check(edgpu_sector_create(d, Hhandle), "build_Hv_sector_gpu")
    call edgpu_check(edgpu_sector_info(Hhandle, Hinfo), "build_Hv_sector_gpu")
    Hstatus = .true.
  end subroutine build_Hv_sector_gpu

  !> delete_Hv_sector(): ED_HAMILTONIAN.f90:174-222
  subroutine delete_Hv_sector_gpu()
    integer(c_int) :: rc
    if (.not. Hstatus) return
    rc = edgpu_sector_destroy(Hhandle)
    Hhandle = c_null_ptr
    Hstatus = .false.
  end subroutine delete_Hv_sector_gpu

  !> vecDim_Hv_sector(isector): ED_HAMILTONIAN.f90:229-253 (default-kind result as in the
  !  reference; use getDim_Hv_sector_gpu for sectors beyond 2^31)
  function vecDim_Hv_sector_gpu() result(vecDim)
    integer :: vecDim
    if (.not. Hstatus) stop "vecDim_Hv_sector_gpu ERROR: Hsector NOT set"
    if (Hinfo%vecDim > int(huge(1), c_int64_t)) stop "vecDim_Hv_sector_gpu ERROR: vecDim exceeds default integer"
    vecDim = int(Hinfo%vecDim)
  end function vecDim_Hv_sector_gpu

  function getDim_Hv_sector_gpu() result(dim)
    integer(8) :: dim
    dim = Hinfo%Dim
  end function getDim_Hv_sector_gpu

  subroutine get_sector_map_gpu(k, map)
    integer, intent(in)  :: k              ! 1..2*Ns_Ud as Hs(k)%map
    integer, intent(out) :: map(:)
    call edgpu_check(edgpu_sector_map(Hhandle, int(k-1, c_int32_t), map), "get_sector_map_gpu")
  end subroutine get_sector_map_gpu

  !> dd_sparse_HxV: bind with  spHtimesV_p => spMatVec_gpu   (ED_HAMILTONIAN.f90:139-166)
  subroutine spMatVec_gpu(Nloc, v, Hv)
    integer                 :: Nloc
    real(8), dimension(Nloc) :: v
    real(8), dimension(Nloc) :: Hv
    if (.not. Hstatus) stop "spMatVec_gpu ERROR: Hsector NOT set"
    call edgpu_check(edgpu_hxv(Hhandle, int(Nloc, c_int32_t), v, Hv), "spMatVec_gpu")
  end subroutine spMatVec_gpu

  !> sp_lanc_eigh(MatVec,egs,vect,Nitermax,iverbose,threshold) with the operator implied
  subroutine sp_lanc_eigh_gpu(egs, vect, Nitermax, threshold, ncheck, nlanc)
    real(8), intent(out)           :: egs
    real(8), intent(inout)         :: vect(:)
    integer, intent(in)            :: Nitermax
    real(8), intent(in), optional  :: threshold
    integer, intent(in), optional  :: ncheck
    integer, intent(out), optional :: nlanc
    real(c_double)     :: thr
    integer(c_int32_t) :: nchk, nl
    thr = 1d-12; if (present(threshold)) thr = threshold
    nchk = 10;   if (present(ncheck)) nchk = ncheck
    if (size(vect) /= Hinfo%vecDim) stop "sp_lanc_eigh_gpu ERROR: size(vect) != vecDim"
    call edgpu_check(edgpu_lanczos_gs(Hhandle, int(Nitermax, c_int32_t), thr, nchk, egs, vect, nl), "sp_lanc_eigh_gpu")
    if (present(nlanc)) nlanc = nl
  end subroutine sp_lanc_eigh_gpu

  !> sp_eigh(MatVec,eval,evec,Nblock,Nitermax,tol,iverbose): ED_DIAG.f90:153-170 (lanc_method default)
  subroutine sp_eigh_gpu(eval, evec, Nblock, Nitermax, tol, nconv)
    real(8), intent(out)           :: eval(:)
    real(8), intent(out)           :: evec(:,:)        ! (vecDim, Neigen)
    integer, intent(in)            :: Nblock, Nitermax
    real(8), intent(in), optional  :: tol
    integer, intent(out), optional :: nconv
    real(c_double)     :: tl
    integer(c_int32_t) :: nc, nr, nm
    tl = 0d0; if (present(tol)) tl = tol
    if (size(evec,1) /= Hinfo%vecDim .or. size(evec,2) /= size(eval)) stop "sp_eigh_gpu ERROR: wrong shape of evec"
    call edgpu_check(edgpu_eigsh(Hhandle, int(size(eval), c_int32_t), int(Nblock, c_int32_t), int(Nitermax, c_int32_t), tl, &
         eval, evec, nc, nr, nm), "sp_eigh_gpu")
    if (present(nconv)) nconv = nc
  end subroutine sp_eigh_gpu

  !> sp_lanc_tridiag(MatVec,vin,alanc,blanc)
  subroutine sp_lanc_tridiag_gpu(vin, alanc, blanc, threshold, nused)
    real(8), intent(in)            :: vin(:)
    real(8), intent(out)           :: alanc(:), blanc(:)
    real(8), intent(in), optional  :: threshold
    integer, intent(out), optional :: nused
    real(c_double)     :: thr
    integer(c_int32_t) :: nu
    thr = 1d-12; if (present(threshold)) thr = threshold
    if (size(alanc) /= size(blanc)) stop "sp_lanc_tridiag_gpu ERROR: size(alanc) != size(blanc)"
    call edgpu_check(edgpu_lanczos_tridiag(Hhandle, vin, int(size(alanc), c_int32_t), thr, alanc, blanc, nu), "sp_lanc_tridiag_gpu")
    if (present(nused)) nused = nu
  end subroutine sp_lanc_tridiag_gpu

end module ED_HAMILTONIAN_GPU
