!> Same-name replacement of the reference module ED_SPARSE_MATRIX (ED_SPARSE_MATRIX.f90:95-105): the public type and
!  generic names a host module USEs -- sparse_matrix_csr, sp_init_matrix, sp_delete_matrix, sp_insert_element,
!  sp_load_matrix, sp_dump_matrix, sp_spy_matrix, sp_set_mpi_matrix, each also in the MpiComm-first form -- directly on
!  the C-ABI container of include/edgpu.h (edgpu_sp_*).  Storage lives behind a handle: code that walks
!  sparse%row(i)%vals itself (only ED_HAMILTONIAN_SPARSE_HxV does, and that module is replaced too) uses sp_matvec /
!  sp_dump_matrix instead.  Compile this file INSTEAD of the reference's ED_SPARSE_MATRIX.f90.
module ED_SPARSE_MATRIX
  use, intrinsic :: iso_c_binding
#ifdef _MPI
  use MPI, only: MPI_COMM_NULL          ! as the reference's module does (ED_SPARSE_MATRIX.f90:2-5)
#endif
  use EDGPU_C_API
  use EDGPU_LIVE, only: edgpu_live_invalidate
  implicit none
  private

  type sparse_matrix_csr
     type(c_ptr) :: h = c_null_ptr
     integer     :: Nrow = 0, Ncol = 0
     logical     :: status = .false.
     integer     :: istart = 0, iend = 0, ishift = 0       ! as in the reference type (ED_SPARSE_MATRIX.f90:22-30)
     logical     :: mpi = .false.
     integer     :: Nglob = 0                               ! rows of the whole matrix (= Nrow unless mpi)
  end type sparse_matrix_csr

  abstract interface
     !> in-place sum of buf(1:n) over the ranks of MpiComm (the host owns MPI): AllReduce_MPI of ED_SPARSE_MATRIX.f90:434
     subroutine sp_allreduce_interface(MpiComm, buf, n)
       integer :: MpiComm, n
       real(8) :: buf(n)
     end subroutine sp_allreduce_interface
  end interface

#ifdef _MPI
  integer, save :: comm_null = MPI_COMM_NULL   !< calls on the null communicator return at once (ED_SPARSE_MATRIX.f90:160,...)
#else
  integer, save :: comm_null = -1     !< value of MPI_COMM_NULL of the host's MPI (sp_set_comm_null) when built without -D_MPI
#endif
  procedure(sp_allreduce_interface), pointer, save :: allreduce_hook => null()

  interface sp_init_matrix
     module procedure :: sp_init_matrix_csr, mpi_sp_init_matrix_csr
  end interface
  interface sp_delete_matrix
     module procedure :: sp_delete_matrix_csr, mpi_sp_delete_matrix_csr
  end interface
  interface sp_insert_element
     module procedure :: sp_insert_element_csr, mpi_sp_insert_element_csr
  end interface
  interface sp_load_matrix
     module procedure :: sp_load_matrix_csr, mpi_sp_load_matrix_csr
  end interface
  interface sp_dump_matrix
     module procedure :: sp_dump_matrix_csr, mpi_sp_dump_matrix_csr
  end interface
  interface sp_spy_matrix
     module procedure :: sp_spy_matrix_csr, mpi_sp_spy_matrix_csr
  end interface
  interface sp_set_mpi_matrix
     module procedure :: sp_set_mpi_matrix_csr
  end interface

  public :: sparse_matrix_csr
  public :: sp_init_matrix, sp_delete_matrix, sp_insert_element, sp_load_matrix, sp_dump_matrix, sp_spy_matrix
  public :: sp_set_mpi_matrix
  public :: sp_matvec            ! extension: the product, on the GPU
  public :: sp_set_comm_null     ! extension: tell the module the host's MPI_COMM_NULL
  public :: sp_set_allreduce     ! extension: the reduction mpi_sp_dump_matrix needs (the host owns MPI)
  public :: sp_allreduce_interface

contains

  subroutine sp_set_comm_null(value)
    integer, intent(in) :: value
    comm_null = value
  end subroutine sp_set_comm_null

  subroutine sp_set_allreduce(hook)
    procedure(sp_allreduce_interface) :: hook
    allreduce_hook => hook
  end subroutine sp_set_allreduce

  subroutine sp_init_matrix_csr(sparse, N, N1)                      ! ED_SPARSE_MATRIX.f90:127-150
    type(sparse_matrix_csr), intent(inout) :: sparse
    integer                                :: N
    integer, optional                      :: N1
    if (sparse%status) stop "sp_init_matrix: already allocated can not init"
    sparse%Nrow = N; sparse%Ncol = N; sparse%Nglob = N
    if (present(N1)) sparse%Ncol = N1
    call edgpu_check(edgpu_sp_init_matrix(int(sparse%Nrow, c_int64_t), int(sparse%Ncol, c_int64_t), sparse%h), "sp_init_matrix")
    sparse%status = .true.
  end subroutine sp_init_matrix_csr

  subroutine mpi_sp_init_matrix_csr(MpiComm, sparse, N, N1)         ! 155-172: holds the rows istart..iend
    integer                                :: MpiComm
    type(sparse_matrix_csr), intent(inout) :: sparse
    integer                                :: N
    integer, optional                      :: N1
    if (MpiComm == comm_null) return
    if (.not. sparse%mpi) stop "mpi_sp_init_matrix_csr: mpi no set"
    if (sparse%status) stop "sp_init_matrix: already allocated can not init"
    sparse%Nglob = N; sparse%Ncol = N
    if (present(N1)) sparse%Ncol = N1
    sparse%Nrow = sparse%iend - sparse%istart + 1
    call edgpu_check(edgpu_sp_init_matrix_mpi(int(sparse%istart, c_int64_t), int(sparse%iend, c_int64_t), int(sparse%ishift, c_int64_t), &
         int(N, c_int64_t), int(sparse%Ncol, c_int64_t), sparse%h), "sp_init_matrix")
    sparse%status = .true.
  end subroutine mpi_sp_init_matrix_csr

  !> 184-237.  The reference calls sp_delete_matrix from delete_Hv_sector only (ED_HAMILTONIAN.f90:186-205) -- it is the
  !  one signal an unchanged host gives that the operator of build_Hv_sector is gone, so the live GPU sector goes with it.
  subroutine sp_delete_matrix_csr(sparse)
    type(sparse_matrix_csr), intent(inout) :: sparse
    integer(c_int) :: rc
    call edgpu_live_invalidate()
    if (.not. sparse%status) return
    rc = edgpu_sp_delete_matrix(sparse%h)
    sparse%h = c_null_ptr; sparse%Nrow = 0; sparse%Ncol = 0; sparse%Nglob = 0; sparse%status = .false.
    sparse%istart = 0; sparse%iend = 0; sparse%ishift = 0; sparse%mpi = .false.
  end subroutine sp_delete_matrix_csr

  subroutine mpi_sp_delete_matrix_csr(MpiComm, sparse)
    integer                                :: MpiComm
    type(sparse_matrix_csr), intent(inout) :: sparse
    if (MpiComm == comm_null) return
    call sp_delete_matrix_csr(sparse)
  end subroutine mpi_sp_delete_matrix_csr

  subroutine sp_insert_element_csr(sparse, value, i, j)             ! 255-288: duplicate-summing append
    type(sparse_matrix_csr), intent(inout) :: sparse
    real(8), intent(in)                    :: value
    integer, intent(in)                    :: i, j
    call edgpu_check(edgpu_sp_insert_element(sparse%h, value, int(i, c_int64_t), int(j, c_int64_t)), "sp_insert_element")
  end subroutine sp_insert_element_csr

  subroutine mpi_sp_insert_element_csr(MpiComm, sparse, value, i, j)   ! 289-323: i is the GLOBAL row, stored at i-ishift
    integer                                :: MpiComm
    type(sparse_matrix_csr), intent(inout) :: sparse
    real(8), intent(in)                    :: value
    integer, intent(in)                    :: i, j
    if (MpiComm == comm_null) return
    if (.not. sparse%mpi) stop "mpi_sp_insert_element_csr: mpi no set"
    call sp_insert_element_csr(sparse, value, i, j)
  end subroutine mpi_sp_insert_element_csr

  subroutine sp_load_matrix_csr(matrix, sparse)                     ! 325-344
    real(8), dimension(:,:), intent(in)    :: matrix
    type(sparse_matrix_csr), intent(inout) :: sparse
    real(8), allocatable :: m(:,:)
    m = matrix                                          ! contiguous copy for the C side
    if (.not. sparse%status) sparse%h = c_null_ptr
    call edgpu_check(edgpu_sp_load_matrix(m, int(size(m,1), c_int64_t), int(size(m,2), c_int64_t), sparse%h), "sp_load_matrix")
    sparse%Nrow = size(m,1); sparse%Ncol = size(m,2); sparse%Nglob = sparse%Nrow; sparse%mpi = .false.; sparse%status = .true.
  end subroutine sp_load_matrix_csr

  subroutine mpi_sp_load_matrix_csr(MpiComm, matrix, sparse)        ! 346-384: only the rows of this rank's window
    integer                                :: MpiComm
    real(8), dimension(:,:), intent(in)    :: matrix
    type(sparse_matrix_csr), intent(inout) :: sparse
    integer :: i, j, ist, ien, ish
    integer(c_int) :: rc
    if (MpiComm == comm_null) return
    if (.not. sparse%mpi) stop "mpi_sp_load_matrix_csr: mpi no set"
    ist = sparse%istart; ien = sparse%iend; ish = sparse%ishift
    if (sparse%status) then
       rc = edgpu_sp_delete_matrix(sparse%h); sparse%h = c_null_ptr; sparse%status = .false.
    end if
    sparse%istart = ist; sparse%iend = ien; sparse%ishift = ish; sparse%mpi = .true.
    call mpi_sp_init_matrix_csr(MpiComm, sparse, size(matrix,1), size(matrix,2))
    do i = ist, ien
       do j = 1, size(matrix,2)
          if (matrix(i,j) /= 0d0) call sp_insert_element_csr(sparse, matrix(i,j), i, j)
       end do
    end do
  end subroutine mpi_sp_load_matrix_csr

  subroutine sp_dump_matrix_csr(sparse, matrix)                     ! 386-406: ADDS the stored entries into matrix
    type(sparse_matrix_csr), intent(in)      :: sparse
    real(8), dimension(:,:), intent(inout)   :: matrix
    real(8), allocatable :: m(:,:)
    if (size(matrix,1) /= sparse%Nglob .or. size(matrix,2) /= sparse%Ncol) stop "Warning SPARSE/dump_matrix: dimensions error"
    m = matrix
    call edgpu_check(edgpu_sp_dump_matrix(sparse%h, m), "sp_dump_matrix")
    matrix = m
  end subroutine sp_dump_matrix_csr

  !> 408-437: the reference zeroes a temporary, dumps this rank's rows into it at their global position, sums it over
  !  MpiComm and OVERWRITES matrix with the result, so every rank ends with the full dense matrix (the LAPACK path of
  !  small sectors depends on it, ED_HAMILTONIAN_SPARSE_HxV.f90:138-182).  The reduction is the host's (it owns MPI):
  !  install it with sp_set_allreduce; without one a matrix split over ranks cannot be dumped correctly and the call stops.
  subroutine mpi_sp_dump_matrix_csr(MpiComm, sparse, matrix)
    integer                                  :: MpiComm
    type(sparse_matrix_csr), intent(in)      :: sparse
    real(8), dimension(:,:), intent(inout)   :: matrix
    real(8), allocatable :: m(:,:)
    if (MpiComm == comm_null) return
    if (.not. sparse%mpi) stop "mpi_sp_dump_matrix_csr: mpi no set"
    if (size(matrix,1) /= sparse%Nglob .or. size(matrix,2) /= sparse%Ncol) stop "Warning SPARSE/mpi_dump_matrix: dimensions error"
    allocate(m(size(matrix,1), size(matrix,2))); m = 0d0
    call edgpu_check(edgpu_sp_dump_matrix(sparse%h, m), "sp_dump_matrix")
    if (sparse%Nrow /= sparse%Nglob) then
       if (.not. associated(allreduce_hook)) stop "mpi_sp_dump_matrix_csr: matrix rows are split over ranks and no reduction is installed (sp_set_allreduce)"
       call allreduce_hook(MpiComm, m, size(m))
    else if (associated(allreduce_hook)) then
       call allreduce_hook(MpiComm, m, size(m))
    end if
    matrix = m
  end subroutine mpi_sp_dump_matrix_csr

  !> row i (global numbering): its stored columns and values in insertion order
  subroutine sp_row(sparse, i, n, cols, vals)
    type(sparse_matrix_csr), intent(in)                   :: sparse
    integer, intent(in)                                   :: i
    integer, intent(out)                                  :: n
    integer(c_int64_t), allocatable, intent(out), target  :: cols(:)
    real(8), allocatable, intent(out), target             :: vals(:)
    integer(c_int32_t) :: sz
    call edgpu_check(edgpu_sp_row(sparse%h, int(i, c_int64_t), sz, c_null_ptr, c_null_ptr, 0_c_int32_t), "sp_row")
    n = sz
    allocate(cols(max(n,1)), vals(max(n,1)))
    if (n > 0) call edgpu_check(edgpu_sp_row(sparse%h, int(i, c_int64_t), sz, c_loc(cols), c_loc(vals), sz), "sp_row")
  end subroutine sp_row

  !> sparsity picture for gnuplot: "<header>_data.dat" holds one "column row" pair per stored entry,
  !  "plot_<header>_commands.gp" draws it (file names as the reference's sp_spy_matrix writes them)
  subroutine sp_spy_matrix_csr(sparse, header)
    type(sparse_matrix_csr)   :: sparse
    character(len=*)          :: header
    integer :: du, cu, i, k, nz, n
    integer(c_int64_t), allocatable :: cols(:)
    real(8), allocatable :: vals(:)
    open(newunit=du, file=trim(header)//"_data.dat", status="replace")
    nz = 0
    do i = 1 + sparse%ishift, sparse%Nrow + sparse%ishift
       call sp_row(sparse, i, n, cols, vals)
       do k = 1, n
          write(du, "(2x,i8,2x,i8)") cols(k), i
       end do
       nz = nz + n
    end do
    close(du)
    open(newunit=cu, file="plot_"//trim(header)//"_commands.gp", status="replace")
    write(cu, "(a)") "set size ratio -1"
    write(cu, "(a)") 'set xlabel "column"'
    write(cu, "(a)") 'set ylabel "row"'
    write(cu, "(a,i0,a)") 'set title "', nz, " stored entries of "//trim(header)//'"'
    write(cu, "(a,i0,a,i0,a)") "plot [1:", sparse%Ncol, "] [", sparse%Nrow + sparse%ishift, ':1] "'//trim(header)//'_data.dat" with points pt 5 ps 0.4'
    close(cu)
  end subroutine sp_spy_matrix_csr

  subroutine mpi_sp_spy_matrix_csr(MpiComm, sparse, header)
    integer                   :: MpiComm
    type(sparse_matrix_csr)   :: sparse
    character(len=*)          :: header
    if (MpiComm == comm_null) return
    call sp_spy_matrix_csr(sparse, header)      ! this rank's rows (the reference writes one file per rank as well)
  end subroutine mpi_sp_spy_matrix_csr

  subroutine sp_set_mpi_matrix_csr(MpiComm, sparse, istart, iend, ishift)   ! 574-585
    integer                                :: MpiComm
    type(sparse_matrix_csr), intent(inout) :: sparse
    integer                                :: istart, iend, ishift
    if (MpiComm == comm_null) return
    sparse%istart = istart; sparse%iend = iend; sparse%ishift = ishift; sparse%mpi = .true.
  end subroutine sp_set_mpi_matrix_csr

  !> Hv = A v (accumulate absent/false) or Hv = Hv + A v, on the device
  subroutine sp_matvec(sparse, v, Hv, accumulate)
    type(sparse_matrix_csr), intent(in) :: sparse
    real(8), intent(in)                 :: v(:)
    real(8), intent(inout)              :: Hv(:)
    logical, intent(in), optional       :: accumulate
    integer(c_int32_t) :: acc
    acc = 0
    if (present(accumulate)) acc = merge(1, 0, accumulate)
    call edgpu_check(edgpu_sp_matvec(sparse%h, v, Hv, acc), "sp_matvec")
  end subroutine sp_matvec

end module ED_SPARSE_MATRIX
