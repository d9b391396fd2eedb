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
  public :: sp_dvdson_eigh_gpu
  public :: get_sector_map_gpu
  ! a second, independently owned sector (the source sector of a Green's-function leg: the reference keeps its
  ! maps in a local `HI` next to the global target sector, ED_GF_NORMAL.f90:176-186)
  public :: sector_gpu, create_sector_gpu, delete_sector_gpu
  ! communicator of the live sector (MpiComm of ed_set_MpiComm, ED_VARS_GLOBAL.f90:332-347)
  public :: get_rccl_unique_id_gpu, set_MpiComm_rccl_gpu, set_MpiComm_callbacks_gpu
  ! callers either side of the product, vectors stay in HBM for the whole leg
  public :: lanc_gf_leg_gpu          ! one leg of lanc_build_gf_normal_main  (ED_GF_NORMAL.f90:184-239 / 248-311)
  public :: lanc_chi_leg_gpu         ! lanc_ed_build_spinChi/densChi legs    (ED_GF_CHISPIN.f90:163-199, ED_GF_CHIDENS.f90:160-197)
  public :: lanc_phonon_leg_gpu      ! lanc_build_gf_phonon_main             (ED_GF_PHONON.f90:40-123)
  public :: imp_histogram_gpu        ! impurity-configuration weights -> every local observable (ED_OBSERVABLES.f90:150-215)
  public :: imp_density_offdiag_gpu  ! ED_OBSERVABLES.f90:283-313
  public :: lanc_local_energy_gpu    ! per-state sums of lanc_local_energy (ED_OBSERVABLES.f90:372-616)
  public :: twin_reorder_gpu         ! es_return_cvector, itwin branch (ED_EIGENSPACE.f90:486-497)

  type sector_gpu
     type(c_ptr)      :: h = c_null_ptr
     type(edgpu_info) :: info
     logical          :: status = .false.
  end type sector_gpu

contains


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
    type(sector_gpu) :: S
    if (Hstatus) stop "build_Hv_sector_gpu ERROR: a sector is already built (call delete_Hv_sector_gpu)"
    call create_sector_gpu(S, Norb, Nspin, Nbath, Nph, bath_type, ed_total_ud, ed_sparse_H, hfmode, &
         Nups, Ndws, Uloc, Ust, Jh, Jx, Jp, xmu, w0_ph, g_ph, impHloc, diag_hybr, bath_diag, hbath_tmp, &
         MpiRank, MpiSize, explicit_csr)
    Hhandle = S%h; Hinfo = S%info
    Hstatus = .true.
  end subroutine build_Hv_sector_gpu

  !> the same constructor for an independently owned sector object
  subroutine create_sector_gpu(S, Norb, Nspin, Nbath, Nph, bath_type, ed_total_ud, ed_sparse_H, hfmode, &
       Nups, Ndws, Uloc, Ust, Jh, Jx, Jp, xmu, w0_ph, g_ph, impHloc, diag_hybr, bath_diag, hbath_tmp, &
       MpiRank, MpiSize, explicit_csr)
    type(sector_gpu), intent(inout)       :: S
    integer, intent(in)                   :: Norb, Nspin, Nbath, Nph
    character(len=*), intent(in)          :: bath_type
    logical, intent(in)                   :: ed_total_ud, ed_sparse_H, hfmode
    integer, intent(in)                   :: Nups(:), Ndws(:)
    real(8), intent(in)                   :: Uloc(:), Ust, Jh, Jx, Jp, xmu, w0_ph, g_ph(:)
    real(8), intent(in), target, contiguous :: impHloc(:,:,:,:)
    real(8), intent(in), target, contiguous :: diag_hybr(:,:,:)
    real(8), intent(in), target, contiguous :: bath_diag(:,:,:)
    real(8), intent(in), target, contiguous, optional :: hbath_tmp(:,:,:,:,:)
    integer, intent(in), optional         :: MpiRank, MpiSize
    logical, intent(in), optional         :: explicit_csr
    type(edgpu_desc) :: d
    integer :: n
    if (S%status) stop "create_sector_gpu ERROR: sector object already in use"
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
    call edgpu_check(edgpu_sector_create(d, S%h), "create_sector_gpu")
    call edgpu_check(edgpu_sector_info(S%h, S%info), "create_sector_gpu")
    S%status = .true.
  end subroutine create_sector_gpu

  subroutine delete_sector_gpu(S)
    type(sector_gpu), intent(inout) :: S
    integer(c_int) :: rc
    if (.not. S%status) return
    rc = edgpu_sector_destroy(S%h)
    S%h = c_null_ptr; S%status = .false.
  end subroutine delete_sector_gpu

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

  !> sp_dvdson_eigh(MatVec,eval,evec,nitermax,tol): ED_DIAG.f90:189-203 (lanc_method="dvdson")
  subroutine sp_dvdson_eigh_gpu(eval, evec, Nitermax, tol, nconv)
    real(8), intent(out)           :: eval(:)
    real(8), intent(out)           :: evec(:,:)        ! (vecDim, Neigen)
    integer, intent(in)            :: Nitermax
    real(8), intent(in), optional  :: tol
    integer, intent(out), optional :: nconv
    real(c_double)     :: tl
    integer(c_int32_t) :: nc, nm
    tl = 0d0; if (present(tol)) tl = tol
    if (size(evec,1) /= Hinfo%vecDim .or. size(evec,2) /= size(eval)) stop "sp_dvdson_eigh_gpu ERROR: wrong shape of evec"
    call edgpu_check(edgpu_dvdson(Hhandle, int(size(eval), c_int32_t), int(Nitermax, c_int32_t), tl, eval, evec, nc, nm), "sp_dvdson_eigh_gpu")
    if (present(nconv)) nconv = nc
  end subroutine sp_dvdson_eigh_gpu

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

  !> RCCL communicator of the live sector: rank 0 calls get_rccl_unique_id_gpu, the host broadcasts the 128 bytes
  !  over MpiComm (MPI_Bcast(id,128,MPI_BYTE,0,MpiComm)), every rank calls set_MpiComm_rccl_gpu.
  subroutine get_rccl_unique_id_gpu(id)
    character(kind=c_char), intent(out) :: id(128)
    call edgpu_check(edgpu_comm_unique_id(id), "get_rccl_unique_id_gpu")
  end subroutine get_rccl_unique_id_gpu
  subroutine set_MpiComm_rccl_gpu(id, S)
    character(kind=c_char), intent(in)       :: id(128)
    type(sector_gpu), intent(in), optional   :: S          ! default: the live sector
    if (present(S)) then
       call edgpu_check(edgpu_comm_init_rccl(S%h, id), "set_MpiComm_rccl_gpu")
    else
       if (.not. Hstatus) stop "set_MpiComm_rccl_gpu ERROR: Hsector NOT set"
       call edgpu_check(edgpu_comm_init_rccl(Hhandle, id), "set_MpiComm_rccl_gpu")
    end if
  end subroutine set_MpiComm_rccl_gpu
  !> or keep MPI: three bind(C) procedures wrapping MPI_Allgatherv / MPI_Alltoallv / MPI_Allreduce (include/edgpu.h)
  subroutine set_MpiComm_callbacks_gpu(allgatherv, exchange, allreduce, ctx, S)
    type(c_funptr), intent(in)             :: allgatherv, exchange, allreduce
    type(c_ptr), intent(in)                :: ctx
    type(sector_gpu), intent(in), optional :: S
    if (present(S)) then
       call edgpu_check(edgpu_comm_init_callbacks(S%h, allgatherv, exchange, allreduce, ctx), "set_MpiComm_callbacks_gpu")
    else
       if (.not. Hstatus) stop "set_MpiComm_callbacks_gpu ERROR: Hsector NOT set"
       call edgpu_check(edgpu_comm_init_callbacks(Hhandle, allgatherv, exchange, allreduce, ctx), "set_MpiComm_callbacks_gpu")
    end if
  end subroutine set_MpiComm_callbacks_gpu

  !> One leg of lanc_build_gf_normal_main: vvinit = c^+_{iorb,ispin}|state> (dagger) or c_{iorb,ispin}|state>,
  !  norm2 = <vvinit|vvinit>, then sp_lanc_tridiag in the LIVE sector (= the target, jsector, as in the reference
  !  which calls build_Hv_sector(jsector), ED_GF_NORMAL.f90:215).  Ssrc is the sector `state` lives in.
  subroutine lanc_gf_leg_gpu(Ssrc, iorb, ispin, dagger, state, alanc, blanc, norm2, nused, threshold)
    type(sector_gpu), intent(in)   :: Ssrc
    integer, intent(in)            :: iorb, ispin
    logical, intent(in)            :: dagger
    real(8), intent(in)            :: state(:)
    real(8), intent(out)           :: alanc(:), blanc(:), norm2
    integer, intent(out), optional :: nused
    real(8), intent(in), optional  :: threshold
    real(c_double)     :: thr
    integer(c_int32_t) :: nu
    if (.not. Hstatus) stop "lanc_gf_leg_gpu ERROR: Hsector NOT set"
    if (size(state) /= Ssrc%info%vecDim) stop "lanc_gf_leg_gpu ERROR: size(state) != vecDim of its sector"
    thr = 1d-12; if (present(threshold)) thr = threshold
    call edgpu_check(edgpu_gf_lanczos(Ssrc%h, Hhandle, int(iorb, c_int32_t), int(ispin, c_int32_t), merge(1_c_int32_t, 0_c_int32_t, dagger), &
         state, int(size(alanc), c_int32_t), thr, norm2, alanc, blanc, nu), "lanc_gf_leg_gpu")
    if (present(nused)) nused = nu
  end subroutine lanc_gf_leg_gpu

  !> Susceptibility leg in the live sector: vvinit = (sum_a cup(a) n_up,a + cdw(a) n_dw,a)|state>
  !  (Sz_a: cup=+1/2, cdw=-1/2; n_a: 1,1), normalised, sp_lanc_tridiag.
  subroutine lanc_chi_leg_gpu(cup, cdw, state, alanc, blanc, norm2, nused, threshold)
    real(8), intent(in)            :: cup(:), cdw(:), state(:)
    real(8), intent(out)           :: alanc(:), blanc(:), norm2
    integer, intent(out), optional :: nused
    real(8), intent(in), optional  :: threshold
    real(c_double)     :: thr
    integer(c_int32_t) :: nu
    if (.not. Hstatus) stop "lanc_chi_leg_gpu ERROR: Hsector NOT set"
    thr = 1d-12; if (present(threshold)) thr = threshold
    call edgpu_check(edgpu_chi_lanczos(Hhandle, cup, cdw, state, int(size(alanc), c_int32_t), thr, norm2, alanc, blanc, nu), "lanc_chi_leg_gpu")
    if (present(nused)) nused = nu
  end subroutine lanc_chi_leg_gpu

  !> Phonon propagator leg in the live sector: vvinit = (b + b^+)|state>
  subroutine lanc_phonon_leg_gpu(state, alanc, blanc, norm2, nused, threshold)
    real(8), intent(in)            :: state(:)
    real(8), intent(out)           :: alanc(:), blanc(:), norm2
    integer, intent(out), optional :: nused
    real(8), intent(in), optional  :: threshold
    real(c_double)     :: thr
    integer(c_int32_t) :: nu
    if (.not. Hstatus) stop "lanc_phonon_leg_gpu ERROR: Hsector NOT set"
    thr = 1d-12; if (present(threshold)) thr = threshold
    call edgpu_check(edgpu_phonon_lanczos(Hhandle, state, int(size(alanc), c_int32_t), thr, norm2, alanc, blanc, nu), "lanc_phonon_leg_gpu")
    if (present(nused)) nused = nu
  end subroutine lanc_phonon_leg_gpu

  !> hist(1 + mu_imp + 2^Norb*md_imp... ) weights of the impurity configurations of a state of the live sector:
  !  hist(1 + md + nI*mu + nI*nI*iph), nI = 2^Norb (C order of include/edgpu.h), summed over ranks
  subroutine imp_histogram_gpu(vec, hist)
    real(8), intent(in)  :: vec(:)
    real(8), intent(out) :: hist(:)
    if (.not. Hstatus) stop "imp_histogram_gpu ERROR: Hsector NOT set"
    call edgpu_check(edgpu_imp_histogram(Hhandle, vec, 0_c_int32_t, hist), "imp_histogram_gpu")
  end subroutine imp_histogram_gpu

  !> dm(b,a,s) = <c^+_a c_b>_s, a /= b, of a state of the live sector (s = 1 up, 2 dw)
  subroutine imp_density_offdiag_gpu(vec, dm)
    real(8), intent(in)  :: vec(:)
    real(8), intent(out) :: dm(:,:,:)        ! (Norb,Norb,2): the C layout [(s*Norb+a)*Norb+b] seen column-major
    if (.not. Hstatus) stop "imp_density_offdiag_gpu ERROR: Hsector NOT set"
    call edgpu_check(edgpu_imp_density_offdiag(Hhandle, vec, dm), "imp_density_offdiag_gpu")
  end subroutine imp_density_offdiag_gpu

  !> ed_Epot (incl. Hartree shift), ed_Ehartree, ed_Eknot, ed_Dust, ed_Dund, ed_Dse, ed_Dph of one state of the live sector,
  !  weight 1: the host multiplies by peso and sums over its state list (ED_OBSERVABLES.f90:391-577)
  subroutine lanc_local_energy_gpu(state, Epot, Ehartree, Eknot, Dust, Dund, Dse, Dph)
    real(8), intent(in)  :: state(:)
    real(8), intent(out) :: Epot, Ehartree, Eknot, Dust, Dund, Dse, Dph
    real(c_double) :: o(7)
    if (.not. Hstatus) stop "lanc_local_energy_gpu ERROR: Hsector NOT set"
    call edgpu_check(edgpu_local_energy(Hhandle, state, o), "lanc_local_energy_gpu")
    Epot = o(1); Ehartree = o(2); Eknot = o(3); Dust = o(4); Dund = o(5); Dse = o(6); Dph = o(7)
  end subroutine lanc_local_energy_gpu

  !> state of the twin sector (ndw,nup) from a state of the live sector
  subroutine twin_reorder_gpu(vin, vout)
    real(8), intent(in)  :: vin(:)
    real(8), intent(out) :: vout(:)
    if (.not. Hstatus) stop "twin_reorder_gpu ERROR: Hsector NOT set"
    call edgpu_check(edgpu_twin_reorder(Hhandle, vin, vout), "twin_reorder_gpu")
  end subroutine twin_reorder_gpu

end module ED_HAMILTONIAN_GPU
