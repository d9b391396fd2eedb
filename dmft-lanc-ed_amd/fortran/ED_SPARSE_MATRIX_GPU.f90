!> Counterpart of ED_SPARSE_MATRIX (ED_SPARSE_MATRIX.f90:95-105) on top of the C ABI: the same
!  procedures on a handle-based type; storage semantics identical (row-wise, duplicate-summing insert
!  in insertion order), the product sp_matvec_gpu runs on the GPU.
module ED_SPARSE_MATRIX_GPU
  use, intrinsic :: iso_c_binding
  use EDGPU_C_API, only: edgpu_error_message
  implicit none
  private

  type sparse_matrix_gpu
     type(c_ptr) :: h = c_null_ptr
     integer     :: Nrow = 0
     integer     :: Ncol = 0
     logical     :: status = .false.
     integer     :: istart = 0, iend = 0, ishift = 0     ! sp_set_mpi_matrix (ED_SPARSE_MATRIX.f90:574-585)
     logical     :: mpi = .false.
     integer     :: Nglob = 0
  end type sparse_matrix_gpu

  interface
     function edgpu_sp_init_matrix(nrow, ncol, out) bind(C, name="edgpu_sp_init_matrix") result(rc)
       import :: c_int, c_int64_t, c_ptr
       integer(c_int64_t), value :: nrow, ncol
       type(c_ptr), intent(out)  :: out
       integer(c_int)            :: rc
     end function edgpu_sp_init_matrix
     function edgpu_sp_init_matrix_mpi(istart, iend, ishift, n, ncol, out) bind(C, name="edgpu_sp_init_matrix_mpi") result(rc)
       import :: c_int, c_int64_t, c_ptr
       integer(c_int64_t), value :: istart, iend, ishift, n, ncol
       type(c_ptr), intent(out)  :: out
       integer(c_int)            :: rc
     end function edgpu_sp_init_matrix_mpi
     function edgpu_sp_load_matrix(dense, nrow, ncol, inout) bind(C, name="edgpu_sp_load_matrix") result(rc)
       import :: c_int, c_int64_t, c_double, c_ptr
       real(c_double), intent(in)  :: dense(*)
       integer(c_int64_t), value   :: nrow, ncol
       type(c_ptr), intent(inout)  :: inout
       integer(c_int)              :: rc
     end function edgpu_sp_load_matrix
     function edgpu_sp_insert_element(m, value, i, j) bind(C, name="edgpu_sp_insert_element") result(rc)
       import :: c_int, c_int64_t, c_double, c_ptr
       type(c_ptr), value        :: m
       real(c_double), value     :: value
       integer(c_int64_t), value :: i, j
       integer(c_int)            :: rc
     end function edgpu_sp_insert_element
     function edgpu_sp_delete_matrix(m) bind(C, name="edgpu_sp_delete_matrix") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: m
       integer(c_int)     :: rc
     end function edgpu_sp_delete_matrix
     function edgpu_sp_dump_matrix(m, dense) bind(C, name="edgpu_sp_dump_matrix") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: m
       real(c_double)     :: dense(*)
       integer(c_int)     :: rc
     end function edgpu_sp_dump_matrix
     function edgpu_sp_row(m, i, size, cols, vals, cap) bind(C, name="edgpu_sp_row") result(rc)
       import :: c_int, c_int32_t, c_int64_t, c_double, c_ptr
       type(c_ptr), value        :: m
       integer(c_int64_t), value :: i
       integer(c_int32_t)        :: size
       type(c_ptr), value        :: cols, vals          ! null: only the size is returned
       integer(c_int32_t), value :: cap
       integer(c_int)            :: rc
     end function edgpu_sp_row
     function edgpu_sp_matvec(m, v, hv, accumulate) bind(C, name="edgpu_sp_matvec") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value         :: m
       real(c_double), intent(in) :: v(*)
       real(c_double)             :: hv(*)
       integer(c_int32_t), value  :: accumulate
       integer(c_int)             :: rc
     end function edgpu_sp_matvec
  end interface

  public :: sparse_matrix_gpu
  public :: sp_init_matrix_gpu, sp_insert_element_gpu, sp_delete_matrix_gpu, sp_dump_matrix_gpu, sp_matvec_gpu
  public :: sp_load_matrix_gpu, sp_set_mpi_matrix_gpu, sp_row_gpu

contains

  subroutine sp_check(rc, where)
    integer(c_int), intent(in)   :: rc
    character(len=*), intent(in) :: where
    if (rc /= 0) then
       write(*,"(A)") trim(where)//": "//edgpu_error_message()
       stop "edgpu error"
    end if
  end subroutine sp_check

  subroutine sp_init_matrix_gpu(sparse, N, N1)            ! ED_SPARSE_MATRIX.f90:127-150
    type(sparse_matrix_gpu), intent(inout) :: sparse
    integer, intent(in)                    :: N
    integer, intent(in), optional          :: N1
    if (sparse%status) stop "sp_init_matrix: already allocated can not init"
    sparse%Nrow = N; sparse%Ncol = N; sparse%Nglob = N
    if (present(N1)) sparse%Ncol = N1
    if (sparse%mpi) then                                   ! mpi_sp_init_matrix: Nloc = iend-istart+1 rows (155-172)
       sparse%Nrow = sparse%iend - sparse%istart + 1
       call sp_check(edgpu_sp_init_matrix_mpi(int(sparse%istart, c_int64_t), int(sparse%iend, c_int64_t), int(sparse%ishift, c_int64_t), &
            int(N, c_int64_t), int(sparse%Ncol, c_int64_t), sparse%h), "sp_init_matrix")
    else
       call sp_check(edgpu_sp_init_matrix(int(sparse%Nrow, c_int64_t), int(sparse%Ncol, c_int64_t), sparse%h), "sp_init_matrix")
    end if
    sparse%status = .true.
  end subroutine sp_init_matrix_gpu

  !> sp_set_mpi_matrix(MpiComm,sparse,istart,iend,ishift): call before sp_init_matrix_gpu; insert/dump then take global rows
  subroutine sp_set_mpi_matrix_gpu(sparse, istart, iend, ishift)   ! ED_SPARSE_MATRIX.f90:574-585
    type(sparse_matrix_gpu), intent(inout) :: sparse
    integer, intent(in)                    :: istart, iend, ishift
    sparse%istart = istart; sparse%iend = iend; sparse%ishift = ishift; sparse%mpi = .true.
  end subroutine sp_set_mpi_matrix_gpu

  !> sp_load_matrix(matrix,sparse): dense -> sparse, an allocated sparse is deleted first   (325-344)
  subroutine sp_load_matrix_gpu(matrix, sparse)
    real(8), intent(in), contiguous        :: matrix(:,:)
    type(sparse_matrix_gpu), intent(inout) :: sparse
    if (.not. sparse%status) sparse%h = c_null_ptr
    call sp_check(edgpu_sp_load_matrix(matrix, int(size(matrix,1), c_int64_t), int(size(matrix,2), c_int64_t), sparse%h), "sp_load_matrix")
    sparse%Nrow = size(matrix,1); sparse%Ncol = size(matrix,2); sparse%Nglob = sparse%Nrow; sparse%mpi = .false.; sparse%status = .true.
  end subroutine sp_load_matrix_gpu

  subroutine sp_insert_element_gpu(sparse, value, i, j)  ! ED_SPARSE_MATRIX.f90:255-288
    type(sparse_matrix_gpu), intent(inout) :: sparse
    real(8), intent(in)                    :: value
    integer, intent(in)                    :: i, j
    call sp_check(edgpu_sp_insert_element(sparse%h, value, int(i, c_int64_t), int(j, c_int64_t)), "sp_insert_element")
  end subroutine sp_insert_element_gpu

  subroutine sp_delete_matrix_gpu(sparse)                ! ED_SPARSE_MATRIX.f90:184-205
    type(sparse_matrix_gpu), intent(inout) :: sparse
    integer(c_int) :: rc
    if (.not. sparse%status) return
    rc = edgpu_sp_delete_matrix(sparse%h)
    sparse%h = c_null_ptr; sparse%Nrow = 0; sparse%Ncol = 0; sparse%status = .false.; sparse%mpi = .false.
  end subroutine sp_delete_matrix_gpu

  subroutine sp_dump_matrix_gpu(sparse, matrix)          ! ED_SPARSE_MATRIX.f90:386-408
    type(sparse_matrix_gpu), intent(in) :: sparse
    real(8), intent(inout), contiguous  :: matrix(:,:)
    if (size(matrix,1) /= sparse%Nglob .or. size(matrix,2) /= sparse%Ncol) stop "Warning SPARSE/dump_matrix: dimensions error"
    call sp_check(edgpu_sp_dump_matrix(sparse%h, matrix), "sp_dump_matrix")
  end subroutine sp_dump_matrix_gpu

  !> row i (global numbering): its stored columns and values in insertion order
  subroutine sp_row_gpu(sparse, i, n, cols, vals)
    type(sparse_matrix_gpu), intent(in)                   :: sparse
    integer, intent(in)                                   :: i
    integer, intent(out)                                  :: n
    integer(c_int64_t), allocatable, intent(out), target  :: cols(:)
    real(8), allocatable, intent(out), target             :: vals(:)
    integer(c_int32_t) :: sz
    call sp_check(edgpu_sp_row(sparse%h, int(i, c_int64_t), sz, c_null_ptr, c_null_ptr, 0_c_int32_t), "sp_row")
    n = sz
    allocate(cols(max(n,1)), vals(max(n,1)))
    if (n > 0) call sp_check(edgpu_sp_row(sparse%h, int(i, c_int64_t), sz, c_loc(cols), c_loc(vals), sz), "sp_row")
  end subroutine sp_row_gpu

  !> Hv = A v (or Hv = Hv + A v when accumulate is true), computed on the device
  subroutine sp_matvec_gpu(sparse, v, Hv, accumulate)
    type(sparse_matrix_gpu), intent(in) :: sparse
    real(8), intent(in)                 :: v(:)
    real(8), intent(inout)              :: Hv(:)
    logical, intent(in), optional       :: accumulate
    integer(c_int32_t) :: acc
    acc = 0
    if (present(accumulate)) acc = merge(1, 0, accumulate)
    call sp_check(edgpu_sp_matvec(sparse%h, v, Hv, acc), "sp_matvec")
  end subroutine sp_matvec_gpu

end module ED_SPARSE_MATRIX_GPU
