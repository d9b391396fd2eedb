!> State shared by the same-name drop-in modules: the ONE live GPU sector (the reference keeps one sector live in the
!  module globals of ED_HAMILTONIAN_COMMON, 11-43) and the process-wide communicator (MpiComm, set once:
!  ED_VARS_GLOBAL.f90:332-347).  No dependence on the host's modules: SF_SP_LINALG's stand-in uses it as well.
module EDGPU_LIVE
  use, intrinsic :: iso_c_binding
  use EDGPU_C_API
  implicit none
  private

  type(c_ptr), save, public      :: live_h = c_null_ptr
  logical, save, public          :: live_status = .false.
  integer, save, public          :: live_isector = 0
  integer(c_int32_t), save, public :: live_storage = -1
  type(edgpu_info), save, public :: live_info
  logical, save, public          :: live_explicit_csr = .false.   !< stored sectors as one explicit CSR instead of the factored form

  type(c_ptr), save              :: world = c_null_ptr            !< communicator over all ranks
  integer, save                  :: world_size = 1, world_rank = 0
  type(c_ptr), save, allocatable :: subs(:)                       !< subs(n): first n ranks (ED_HAMILTONIAN.f90:65-93), n < world_size
  logical, save, allocatable     :: subs_owned(:)                 !< subs(n) was created by edgpu_live_add_sub_callbacks (destroyed here, not with `world`)

  public :: edgpu_live_invalidate, edgpu_live_adopt, edgpu_live_hxv, edgpu_live_require
  public :: edgpu_live_set_comm_rccl, edgpu_live_set_comm_callbacks, edgpu_live_add_sub_callbacks, edgpu_live_del_comm, edgpu_live_attach_comm
  public :: edgpu_live_unique_id

contains

  !> the host deleted its sector (delete_Hv_sector calls sp_delete_matrix, ED_HAMILTONIAN.f90:186-205): free the GPU side
  subroutine edgpu_live_invalidate()
    integer(c_int) :: rc
    if (live_status) rc = edgpu_sector_destroy(live_h)
    live_h = c_null_ptr; live_status = .false.; live_isector = 0; live_storage = -1
  end subroutine edgpu_live_invalidate

  subroutine edgpu_live_adopt(h, isector, storage)
    type(c_ptr), intent(in)        :: h
    integer, intent(in)            :: isector
    integer(c_int32_t), intent(in) :: storage
    call edgpu_live_invalidate()
    live_h = h; live_isector = isector; live_storage = storage; live_status = .true.
    call edgpu_check(edgpu_sector_info(live_h, live_info), "edgpu_live_adopt")
  end subroutine edgpu_live_adopt

  subroutine edgpu_live_require(where)
    character(len=*), intent(in) :: where
    if (.not. live_status) then
       write(*,"(A)") trim(where)//" ERROR: Hsector NOT set (no live GPU sector)"
       stop "edgpu error"
    end if
  end subroutine edgpu_live_require

  !> dd_sparse_HxV on the live sector (host arrays: one PCIe round trip per product -- the correctness path)
  subroutine edgpu_live_hxv(Nloc, v, Hv)
    integer                  :: Nloc
    real(8), dimension(Nloc) :: v, Hv
    call edgpu_live_require("spHtimesV")
    call edgpu_check(edgpu_hxv(live_h, int(Nloc, c_int32_t), v, Hv), "spHtimesV")
  end subroutine edgpu_live_hxv

  subroutine edgpu_live_unique_id(id)
    character(kind=c_char), intent(out) :: id(128)
    call edgpu_check(edgpu_comm_unique_id(id), "edgpu_live_unique_id")
  end subroutine edgpu_live_unique_id

  !> once per process, after ed_set_MpiComm: rank 0 makes the id (edgpu_live_unique_id), the host broadcasts its 128 bytes.
  !  The communicators of the first n ranks (sectors with fewer dw columns than ranks) are split here, while every rank is
  !  present: the ranks such a sector leaves out return from the builders and from the SF_SP_LINALG entry points at once
  !  (MpiComm == MPI_COMM_NULL, as ED_HAMILTONIAN_SPARSE_HxV.f90:35-37) and never create or attach anything.
  subroutine edgpu_live_set_comm_rccl(id, rank, nranks, device)
    character(kind=c_char), intent(in) :: id(128)
    integer, intent(in)                :: rank, nranks
    integer, intent(in), optional      :: device
    integer(c_int32_t) :: dev, mem
    integer :: n
    dev = -1; if (present(device)) dev = int(device, c_int32_t)
    call edgpu_live_del_comm()
    call edgpu_check(edgpu_comm_create_rccl(id, int(rank, c_int32_t), int(nranks, c_int32_t), dev, world), "edgpu_live_set_comm_rccl")
    world_size = nranks; world_rank = rank
    allocate(subs(max(nranks - 1, 1)), subs_owned(max(nranks - 1, 1))); subs = c_null_ptr; subs_owned = .false.
    do n = 1, nranks - 1
       call edgpu_check(edgpu_comm_sub(world, int(n, c_int32_t), subs(n), mem), "edgpu_live_set_comm_rccl")
    end do
  end subroutine edgpu_live_set_comm_rccl

  !> or keep the host's MPI: bind(C) procedures wrapping MPI_Allgatherv / MPI_Alltoallv / MPI_Allreduce on MpiComm_Global.
  !  Sectors with 2 <= DimDw < MpiSize run on the communicator of the first DimDw ranks (ED_HAMILTONIAN.f90:65-93): a host
  !  that wants them on this route registers the callbacks of each such group once with edgpu_live_add_sub_callbacks
  !  (collectives of the world communicator cannot serve a sub-group).
  subroutine edgpu_live_set_comm_callbacks(rank, nranks, allgatherv, exchange, allreduce, ctx)
    integer, intent(in)        :: rank, nranks
    type(c_funptr), intent(in) :: allgatherv, exchange, allreduce
    type(c_ptr), intent(in)    :: ctx
    call edgpu_live_del_comm()
    call edgpu_check(edgpu_comm_create_callbacks(int(rank, c_int32_t), int(nranks, c_int32_t), allgatherv, exchange, allreduce, ctx, world), &
         "edgpu_live_set_comm_callbacks")
    world_size = nranks; world_rank = rank
    allocate(subs(max(nranks - 1, 1)), subs_owned(max(nranks - 1, 1))); subs = c_null_ptr; subs_owned = .false.
  end subroutine edgpu_live_set_comm_callbacks

  !> callbacks route: the collectives of the group of the first `n` ranks (an MPI communicator the host made with
  !  MPI_Comm_create on ranks 0..n-1, passed through `ctx`).  Called by the MEMBERS of that group only (rank < n), after
  !  edgpu_live_set_comm_callbacks; 2 <= n < MpiSize.
  subroutine edgpu_live_add_sub_callbacks(n, allgatherv, exchange, allreduce, ctx)
    integer, intent(in)        :: n
    type(c_funptr), intent(in) :: allgatherv, exchange, allreduce
    type(c_ptr), intent(in)    :: ctx
    integer(c_int) :: rc
    if (.not. c_associated(world)) stop "edgpu_live_add_sub_callbacks: call edgpu_live_set_comm_callbacks first"
    if (n < 2 .or. n >= world_size) stop "edgpu_live_add_sub_callbacks: need 2 <= n < MpiSize"
    if (world_rank >= n) stop "edgpu_live_add_sub_callbacks: this rank is not a member of the first n ranks"
    if (c_associated(subs(n)) .and. subs_owned(n)) rc = edgpu_comm_destroy(subs(n))
    call edgpu_check(edgpu_comm_create_callbacks(int(world_rank, c_int32_t), int(n, c_int32_t), allgatherv, exchange, allreduce, ctx, subs(n)), &
         "edgpu_live_add_sub_callbacks")
    subs_owned(n) = .true.
  end subroutine edgpu_live_add_sub_callbacks

  subroutine edgpu_live_del_comm()
    integer(c_int) :: rc
    integer :: n
    if (allocated(subs)) then
       do n = 1, size(subs)
          if (subs_owned(n) .and. c_associated(subs(n))) rc = edgpu_comm_destroy(subs(n))
       end do
       deallocate(subs, subs_owned)
    end if
    if (c_associated(world)) rc = edgpu_comm_destroy(world)      ! destroys the sub-communicators split from it as well
    world = c_null_ptr; world_size = 1; world_rank = 0
  end subroutine edgpu_live_del_comm

  !> attach the communicator of the first `nranks` ranks to a freshly created sector
  subroutine edgpu_live_attach_comm(h, nranks)
    type(c_ptr), intent(in) :: h
    integer, intent(in)     :: nranks
    if (nranks <= 1) return
    if (.not. c_associated(world)) stop "edgpu: MpiSize>1 but no communicator (call edgpu_live_set_comm_rccl/_callbacks after ed_set_MpiComm)"
    if (nranks == world_size) then
       call edgpu_check(edgpu_sector_attach_comm(h, world), "edgpu_live_attach_comm")
    else
       if (.not. allocated(subs)) stop "edgpu: no communicator set"
       if (.not. c_associated(subs(nranks))) stop "edgpu: no communicator for this shrunk sector (callbacks route: edgpu_live_add_sub_callbacks(n,...); or this rank is outside the first n ranks)"
       call edgpu_check(edgpu_sector_attach_comm(h, subs(nranks)), "edgpu_live_attach_comm")
    end if
  end subroutine edgpu_live_attach_comm

end module EDGPU_LIVE
