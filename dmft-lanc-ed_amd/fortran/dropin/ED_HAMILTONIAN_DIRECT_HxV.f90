!> Same-name replacement of the reference module ED_HAMILTONIAN_DIRECT_HxV (public names of its lines 9-14): the on-the-fly
!  product (ed_sparse_H=F).  Nothing is built by build_Hv_sector in this mode -- it only binds the pointer
!  (ED_HAMILTONIAN.f90:152-165) -- so the GPU sector (storage DIRECT: nothing of size N stored) is created at the first
!  product after a build; delete_Hv_sector's sp_delete_matrix calls end it.
MODULE ED_HAMILTONIAN_DIRECT_HxV
  use, intrinsic :: iso_c_binding
  USE ED_HAMILTONIAN_COMMON
  USE ED_HAMILTONIAN_GPU_BUILD
  use EDGPU_C_API
  use EDGPU_LIVE
  implicit none
  private

  public :: directMatVec_main
  public :: directMatVec_orbs
  public :: directMatVec_MPI_main
  public :: directMatVec_MPI_orbs

contains

  subroutine direct_product(Nloc, vin, Hv, who)
    integer                  :: Nloc
    real(8), dimension(Nloc) :: vin, Hv
    character(len=*)         :: who
#ifdef _MPI
    if (MpiStatus .and. MpiComm == MPI_COMM_NULL) return                      ! a rank outside the shrunk communicator holds no rows
#endif
    if (.not. Hstatus) stop "directMatVec_cc ERROR: Hsector NOT set"          ! reference 30
    if (.not. gpu_live_is_current(EDGPU_STORE_DIRECT)) call gpu_build_live(EDGPU_STORE_DIRECT)
    if (Nloc /= live_info%vecDim) then
       write(*,"(A)") who//" ERROR: Nloc != dim(isector)"                       ! reference 35
       stop "directMatVec"
    end if
    call edgpu_live_hxv(Nloc, vin, Hv)
  end subroutine direct_product

  subroutine directMatVec_main(Nloc, vin, Hv)
    integer                  :: Nloc
    real(8), dimension(Nloc) :: vin, Hv
    call direct_product(Nloc, vin, Hv, "directMatVec_main")
  end subroutine directMatVec_main
  subroutine directMatVec_orbs(Nloc, vin, Hv)
    integer                  :: Nloc
    real(8), dimension(Nloc) :: vin, Hv
    call direct_product(Nloc, vin, Hv, "directMatVec_orbs")
  end subroutine directMatVec_orbs
  subroutine directMatVec_MPI_main(Nloc, vin, Hv)
    integer                  :: Nloc
    real(8), dimension(Nloc) :: vin, Hv
    call direct_product(Nloc, vin, Hv, "directMatVec_MPI_main")
  end subroutine directMatVec_MPI_main
  subroutine directMatVec_MPI_orbs(Nloc, vin, Hv)
    integer                  :: Nloc
    real(8), dimension(Nloc) :: vin, Hv
    call direct_product(Nloc, vin, Hv, "directMatVec_MPI_orbs")
  end subroutine directMatVec_MPI_orbs

END MODULE ED_HAMILTONIAN_DIRECT_HxV
