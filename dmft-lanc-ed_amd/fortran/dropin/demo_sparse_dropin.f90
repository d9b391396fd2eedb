!> The container through the reference's own module and procedure names (USE ED_SPARSE_MATRIX): host-side
!  semantics only, runs without a GPU.
program demo_sparse_dropin
  use ED_SPARSE_MATRIX
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
  F = 0d0; call sp_dump_matrix(comm, W, F)
  write(*,"(A,I3,2F8.3)") "window=", W%Nrow, F(3,4), F(5,1)
  call sp_delete_matrix(comm, W)
end program demo_sparse_dropin
