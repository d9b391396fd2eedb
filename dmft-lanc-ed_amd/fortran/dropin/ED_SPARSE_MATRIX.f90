!> Same-name stand-in for the reference module ED_SPARSE_MATRIX (ED_SPARSE_MATRIX.f90:95-105): the public type
!  and generic names a host module USEs -- sparse_matrix_csr, sp_init_matrix, sp_delete_matrix, sp_insert_element,
!  sp_load_matrix, sp_dump_matrix, sp_spy_matrix, sp_set_mpi_matrix, each also in the MpiComm-first form -- on top
!  of the C-ABI container (ED_SPARSE_MATRIX_GPU).  Storage lives behind a handle: code that walks
!  sparse%row(i)%vals directly (only ED_HAMILTONIAN_SPARSE_HxV does, and that module is replaced) must use
!  sp_matvec / sp_dump_matrix instead.  Link this file INSTEAD of the reference's ED_SPARSE_MATRIX.f90.
module ED_SPARSE_MATRIX
  use, intrinsic :: iso_c_binding
  use ED_SPARSE_MATRIX_GPU
  implicit none
  private

  type sparse_matrix_csr
     type(sparse_matrix_gpu) :: g
     integer :: Nrow = 0, Ncol = 0
     logical :: status = .false.
     integer :: istart = 0, iend = 0, ishift = 0       ! as in the reference type (ED_SPARSE_MATRIX.f90:22-30)
     logical :: mpi = .false.
  end type sparse_matrix_csr

  integer, save :: comm_null = -1     !< value of MPI_COMM_NULL of the host's MPI (sp_set_comm_null); calls on it return at once

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

contains

  subroutine sp_set_comm_null(value)
    integer, intent(in) :: value
    comm_null = value
  end subroutine sp_set_comm_null

  subroutine sync_fields(sparse)
    type(sparse_matrix_csr), intent(inout) :: sparse
    sparse%Nrow = sparse%g%Nrow; sparse%Ncol = sparse%g%Ncol; sparse%status = sparse%g%status
  end subroutine sync_fields

  subroutine sp_init_matrix_csr(sparse, N, N1)
    type(sparse_matrix_csr), intent(inout) :: sparse
    integer                                :: N
    integer, optional                      :: N1
    if (sparse%status) stop "sp_init_matrix: already allocated can not init"
    call sp_init_matrix_gpu(sparse%g, N, N1)
    call sync_fields(sparse)
  end subroutine sp_init_matrix_csr

  subroutine mpi_sp_init_matrix_csr(MpiComm, sparse, N, N1)
    integer                                :: MpiComm
    type(sparse_matrix_csr), intent(inout) :: sparse
    integer                                :: N
    integer, optional                      :: N1
    if (MpiComm == comm_null) return
    if (.not. sparse%mpi) stop "mpi_sp_init_matrix_csr: mpi no set"
    if (sparse%status) stop "sp_init_matrix: already allocated can not init"
    call sp_set_mpi_matrix_gpu(sparse%g, sparse%istart, sparse%iend, sparse%ishift)
    call sp_init_matrix_gpu(sparse%g, N, N1)           ! holds iend-istart+1 rows
    call sync_fields(sparse)
  end subroutine mpi_sp_init_matrix_csr

  subroutine sp_delete_matrix_csr(sparse)
    type(sparse_matrix_csr), intent(inout) :: sparse
    if (.not. sparse%status) return
    call sp_delete_matrix_gpu(sparse%g)
    call sync_fields(sparse)
    sparse%istart = 0; sparse%iend = 0; sparse%ishift = 0; sparse%mpi = .false.
  end subroutine sp_delete_matrix_csr

  subroutine mpi_sp_delete_matrix_csr(MpiComm, sparse)
    integer                                :: MpiComm
    type(sparse_matrix_csr), intent(inout) :: sparse
    if (MpiComm == comm_null) return
    call sp_delete_matrix_csr(sparse)
  end subroutine mpi_sp_delete_matrix_csr

  subroutine sp_insert_element_csr(sparse, value, i, j)
    type(sparse_matrix_csr), intent(inout) :: sparse
    real(8), intent(in)                    :: value
    integer, intent(in)                    :: i, j
    call sp_insert_element_gpu(sparse%g, value, i, j)
  end subroutine sp_insert_element_csr

  subroutine mpi_sp_insert_element_csr(MpiComm, sparse, value, i, j)
    integer                                :: MpiComm
    type(sparse_matrix_csr), intent(inout) :: sparse
    real(8), intent(in)                    :: value
    integer, intent(in)                    :: i, j          ! i: GLOBAL row, stored at i-ishift
    if (MpiComm == comm_null) return
    if (.not. sparse%mpi) stop "mpi_sp_insert_element_csr: mpi no set"
    call sp_insert_element_gpu(sparse%g, value, i, j)
  end subroutine mpi_sp_insert_element_csr

  subroutine sp_load_matrix_csr(matrix, sparse)
    real(8), dimension(:,:), intent(in)    :: matrix
    type(sparse_matrix_csr), intent(inout) :: sparse
    real(8), allocatable :: m(:,:)
    m = matrix                                          ! contiguous copy for the C side
    call sp_load_matrix_gpu(m, sparse%g)
    call sync_fields(sparse)
    sparse%mpi = .false.
  end subroutine sp_load_matrix_csr

  subroutine mpi_sp_load_matrix_csr(MpiComm, matrix, sparse)
    integer                                :: MpiComm
    real(8), dimension(:,:), intent(in)    :: matrix
    type(sparse_matrix_csr), intent(inout) :: sparse
    integer :: i, j, ist, ien, ish
    if (MpiComm == comm_null) return
    if (.not. sparse%mpi) stop "mpi_sp_load_matrix_csr: mpi no set"
    ist = sparse%istart; ien = sparse%iend; ish = sparse%ishift
    if (sparse%status) call sp_delete_matrix_gpu(sparse%g)
    call sp_set_mpi_matrix_gpu(sparse%g, ist, ien, ish)
    call sp_init_matrix_gpu(sparse%g, size(matrix,1), size(matrix,2))
    do i = ist, ien                                     ! only the rows of this rank's window
       do j = 1, size(matrix,2)
          if (matrix(i,j) /= 0d0) call sp_insert_element_gpu(sparse%g, matrix(i,j), i, j)
       end do
    end do
    call sync_fields(sparse)
  end subroutine mpi_sp_load_matrix_csr

  subroutine sp_dump_matrix_csr(sparse, matrix)
    type(sparse_matrix_csr), intent(in)      :: sparse
    real(8), dimension(:,:), intent(inout)   :: matrix
    real(8), allocatable :: m(:,:)
    m = matrix
    call sp_dump_matrix_gpu(sparse%g, m)
    matrix = m
  end subroutine sp_dump_matrix_csr

  !> adds this rank's rows at their global position; the reference then sums the dense array over the communicator
  !  (AllReduce_MPI, ED_SPARSE_MATRIX.f90:434) -- that reduction stays with the host, which owns MPI
  subroutine mpi_sp_dump_matrix_csr(MpiComm, sparse, matrix)
    integer                                  :: MpiComm
    type(sparse_matrix_csr), intent(in)      :: sparse
    real(8), dimension(:,:), intent(inout)   :: matrix
    if (MpiComm == comm_null) return
    call sp_dump_matrix_csr(sparse, matrix)
  end subroutine mpi_sp_dump_matrix_csr

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
       call sp_row_gpu(sparse%g, i, n, cols, vals)
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

  subroutine sp_set_mpi_matrix_csr(MpiComm, sparse, istart, iend, ishift)
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
    call sp_matvec_gpu(sparse%g, v, Hv, accumulate)
  end subroutine sp_matvec

end module ED_SPARSE_MATRIX
