!> Builder shared by the same-name operator modules: turns what the reference holds in ED_INPUT_VARS / ED_VARS_GLOBAL /
!  ED_HAMILTONIAN_COMMON when build_Hv_sector runs (ED_HAMILTONIAN.f90:43-168) into an edgpu_desc and makes the GPU sector
!  the live one.  USEs the host's own modules: compile it inside the host tree, after ED_HAMILTONIAN_COMMON.
module ED_HAMILTONIAN_GPU_BUILD
  use, intrinsic :: iso_c_binding
  USE ED_HAMILTONIAN_COMMON
  use EDGPU_C_API
  use EDGPU_LIVE
  implicit none
  private

  public :: gpu_build_live, gpu_live_is_current

  real(8), allocatable, target, save :: hbath_keep(:,:,:,:,:)     ! (Nspin,Nspin,Norb,Norb,Nbath), replica baths
  real(8), allocatable, target, save :: imp_keep(:,:,:,:), dh_keep(:,:,:), bd_keep(:,:,:)    ! contiguous copies with the TARGET attribute

contains

  !> is the live GPU sector the operator of the sector the host has built (Hsector) in this storage?
  logical function gpu_live_is_current(storage)
    integer(c_int32_t), intent(in) :: storage
    gpu_live_is_current = live_status .and. live_isector == Hsector .and. live_storage == storage
  end function gpu_live_is_current

  !> bath arrays exactly as ed_buildh_main resolves them (ED_HAMILTONIAN_SPARSE_HxV.f90:43-75) and directMatVec_main does per
  !  call (ED_HAMILTONIAN_DIRECT_HxV.f90:37-69): diag_hybr, bath_diag (module variables of ED_HAMILTONIAN_COMMON) and, for
  !  replica baths, the per-bath-site matrices bath_from_sym(lambda)
  subroutine resolve_bath()
    integer :: kb, is, io
    if (allocated(diag_hybr)) deallocate(diag_hybr)
    if (allocated(bath_diag)) deallocate(bath_diag)
    if (allocated(hbath_keep)) deallocate(hbath_keep)
    if (trim(bath_type) == "replica") then
       Nfoo = Norb
       allocate(diag_hybr(Nspin,Norb,Nbath), bath_diag(Nspin,Norb,Nbath), hbath_keep(Nspin,Nspin,Norb,Norb,Nbath))
       do kb = 1, Nbath
          hbath_keep(:,:,:,:,kb) = bath_from_sym(dmft_bath%item(kb)%lambda)
          do is = 1, Nspin
             do io = 1, Norb
                diag_hybr(is,io,kb) = dmft_bath%item(kb)%v(is)
                bath_diag(is,io,kb) = hbath_keep(is,is,io,io,kb)
             end do
          end do
       end do
    else
       Nfoo = size(dmft_bath%e, 2)
       allocate(diag_hybr(Nspin,Norb,Nbath), bath_diag(Nspin,Nfoo,Nbath))
       diag_hybr = dmft_bath%v(1:Nspin,1:Norb,1:Nbath)
       bath_diag = dmft_bath%e(1:Nspin,1:Nfoo,1:Nbath)
    end if
  end subroutine resolve_bath

  !> create the GPU sector of Hsector in `storage` (EDGPU_STORE_*) and make it the live one
  subroutine gpu_build_live(storage)
    integer(c_int32_t), intent(in) :: storage
    type(edgpu_desc) :: d
    type(c_ptr)      :: h
    integer          :: Nups(Ns_Ud), Ndws(Ns_Ud)
    if (.not. Hstatus) stop "gpu_build_live ERROR: Hsector NOT set"
#ifdef _MPI
    if (MpiStatus .and. MpiComm == MPI_COMM_NULL) stop "gpu_build_live ERROR: called on a rank outside the sector's communicator"
#endif
    call resolve_bath()
    call get_Nup(Hsector, Nups)
    call get_Ndw(Hsector, Ndws)
    if (allocated(imp_keep)) deallocate(imp_keep)
    allocate(imp_keep(Nspin,Nspin,Norb,Norb)); imp_keep = impHloc
    if (allocated(dh_keep)) deallocate(dh_keep, bd_keep)
    allocate(dh_keep(Nspin,Norb,Nbath), bd_keep(Nspin,Nfoo,Nbath)); dh_keep = diag_hybr; bd_keep = bath_diag
    d%abi_version = EDGPU_ABI_VERSION
    d%Norb = Norb; d%Nspin = Nspin; d%Nbath = Nbath; d%Nph = DimPh - 1
    select case (trim(bath_type))
    case ("hybrid");  d%bath_type = EDGPU_BATH_HYBRID
    case ("replica"); d%bath_type = EDGPU_BATH_REPLICA
    case default;     d%bath_type = EDGPU_BATH_NORMAL
    end select
    d%ed_total_ud = merge(1, 0, ed_total_ud)
    d%storage = storage
    d%hfmode = merge(1, 0, HFmode)
    d%Nfoo = Nfoo
    d%nups = 0; d%ndws = 0
    d%nups(1:Ns_Ud) = Nups; d%ndws(1:Ns_Ud) = Ndws
    d%mpi_rank = 0; d%mpi_size = 1
    if (MpiStatus) then                      ! build_Hv_sector has already shrunk them for sectors with DimDw < MpiSize (65-93)
       d%mpi_rank = MpiRank; d%mpi_size = MpiSize
    end if
    d%device = -1; d%reserved = 0
    d%Uloc = Uloc; d%Ust = Ust; d%Jh = Jh; d%Jx = Jx; d%Jp = Jp; d%xmu = xmu
    d%w0_ph = w0_ph; d%g_ph = g_ph
    d%impHloc = c_loc(imp_keep); d%diag_hybr = c_loc(dh_keep); d%bath_diag = c_loc(bd_keep)
    d%hbath = c_null_ptr
    if (allocated(hbath_keep)) d%hbath = c_loc(hbath_keep)
    call edgpu_check(edgpu_sector_create(d, h), "build_Hv_sector")
    call edgpu_live_adopt(h, Hsector, storage)
    if (live_info%isector /= Hsector) stop "gpu_build_live ERROR: sector numbering differs from get_Sector"
    if (d%mpi_size > 1) call edgpu_live_attach_comm(h, int(d%mpi_size))
  end subroutine gpu_build_live

end module ED_HAMILTONIAN_GPU_BUILD
