!> TEST SCAFFOLDING ONLY.  What `USE MPI` gives an MPI build of the host, as far as the same-name drop-in modules look at
!  it: the handle of the null communicator (ED_HAMILTONIAN_SPARSE_HxV.f90:36, ED_HAMILTONIAN.f90:89 compare MpiComm with it).
!  No MPI library is present in this image; nothing here communicates.
MODULE MPI
  implicit none
  integer, parameter :: MPI_COMM_NULL = 67108864, MPI_COMM_WORLD = 1140850688
END MODULE MPI
