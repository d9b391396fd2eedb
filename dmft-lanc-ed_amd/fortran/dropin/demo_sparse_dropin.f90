!> The container through the reference's own module and procedure names (USE ED_SPARSE_MATRIX): host-side
!  semantics only, runs without a GPU.
module demo_peer
  implicit none
  real(8) :: peer(6,4) = 0d0          !< what the OTHER rank's rows would contribute to the dense matrix
contains
  !> stands in for AllReduce_MPI over two ranks: this rank's buffer + the peer's
  subroutine fake_allreduce(MpiComm, buf, n)
    integer :: MpiComm, n
    real(8) :: buf(n)
    buf = buf + reshape(peer, [n])
  end subroutine fake_allreduce
end module demo_peer

program demo_sparse_dropin
  use ED_SPARSE_MATRIX
  use demo_peer
  implicit none
  type(sparse_matrix_csr) :: A, B, W
  real(8) :: M(3,4), D(3,4), F(6,4)
  integer, parameter :: comm = 7
  call sp_init_matrix(A, 3, 4)
  call sp_insert_element(A, 2d0, 1, 1); call sp_insert_element(A, -1d0, 1, 3)
  call sp_insert_element(A, 0.5d0, 1, 3); call sp_insert_element(A, 4d0, 3, 2)
  D = 0d0; call sp_dump_matrix(A, D)
  write(*,"(A,4F8.3)") "row1=", D(1,:)
  write(*,"(A,4F8.3)") "row3=", D(3,:)
  write(*,"(A,2I3,L2)") "shape_status=", A%Nrow, A%Ncol, A%status
  M = 0d0; M(2,4) = 7d0; M(2,1) = -3d0
  call sp_load_matrix(M, B)
  D = 0d0; call sp_dump_matrix(B, D)
  write(*,"(A,4F8.3)") "loaded_row2=", D(2,:)
  call sp_spy_matrix(B, "dropinB")
  call sp_delete_matrix(A); call sp_delete_matrix(B)
  write(*,"(A,L2)") "deleted=", A%status
  ! MPI forms: rows 3..5 of a 6 x 4 matrix live here
  call sp_set_mpi_matrix(comm, W, 3, 5, 2)
  call sp_init_matrix(comm, W, 6, 4)
  call sp_insert_element(comm, W, 1.5d0, 3, 4); call sp_insert_element(comm, W, -2d0, 5, 1); call sp_insert_element(comm, W, 0.5d0, 5, 1)
  ! mpi_sp_dump_matrix (ED_SPARSE_MATRIX.f90:408-437): zero a temporary, dump the local rows, SUM OVER THE RANKS, OVERWRITE
  peer = 0d0; peer(1,2) = 9d0; peer(6,3) = -4d0          ! rows 1-2 and 6 live on the other rank
  call sp_set_allreduce(fake_allreduce)
  F = 777d0                                               ! must not survive: the reference overwrites, it does not add
  call sp_dump_matrix(comm, W, F)
  write(*,"(A,I3,2F8.3)") "window=", W%Nrow, F(3,4), F(5,1)
  write(*,"(A,3F8.3)") "reduced=", F(1,2), F(6,3), F(2,2)
  call sp_delete_matrix(comm, W)
end program demo_sparse_dropin
