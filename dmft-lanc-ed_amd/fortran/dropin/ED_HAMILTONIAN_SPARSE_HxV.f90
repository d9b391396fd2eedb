!> Same-name replacement of the reference module ED_HAMILTONIAN_SPARSE_HxV (public names of its lines 9-19): the stored
!  sector operator is built and applied on the GPU through the C ABI of include/edgpu.h.  ED_HAMILTONIAN.f90 binds these
!  names unmodified (139-166):   call ed_buildh_main(isector) ;  spHtimesV_p => spMatVec_main   (or _MPI_ / _orbs).
!  Compile this file INSTEAD of the reference's, together with EDGPU_C_API, EDGPU_LIVE, ED_HAMILTONIAN_GPU_BUILD and the
!  same-name ED_SPARSE_MATRIX / ED_HAMILTONIAN_DIRECT_HxV, with the host's own preprocessor flags (-cpp, -D_MPI for an MPI
!  build: MPI_COMM_NULL then arrives through ED_VARS_GLOBAL's USE MPI as it does for the reference's file).
MODULE ED_HAMILTONIAN_SPARSE_HxV
  use, intrinsic :: iso_c_binding
  USE ED_HAMILTONIAN_COMMON
  USE ED_HAMILTONIAN_GPU_BUILD
  use EDGPU_C_API
  use EDGPU_LIVE
  implicit none
  private

  !>Sparse Matrix constructors
  public :: ed_buildh_main
  public :: ed_buildh_orbs
  !>Sparse Mat-Vec product using the stored operator
  public :: spMatVec_main
  public :: spMatVec_orbs
  public :: spMatVec_MPI_main
  public :: spMatVec_MPI_orbs

contains

  !> ed_buildh_main(isector[,Hmat]) (reference 25-200): the stored operator of Hsector, in HBM; with Hmat, the dense
  !  matrix the LAPACK path of small sectors diagonalises (132-195).  spH0d, spH0ups, ... are not filled: the operator
  !  lives behind the handle (sp_delete_matrix on them in delete_Hv_sector stays a no-op and frees the GPU sector).
  subroutine ed_buildh_main(isector, Hmat)
    integer                           :: isector
    real(8), dimension(:,:), optional :: Hmat
    if (.not. ed_total_ud) stop "ed_buildh_main ERROR: ed_total_ud=F takes ed_buildh_orbs"
    call buildh(isector, Hmat, "ed_buildh_main")
  end subroutine ed_buildh_main

  !> ed_buildh_orbs(isector[,Hmat]) (reference 206-370): ed_total_ud=F
  subroutine ed_buildh_orbs(isector, Hmat)
    integer                           :: isector
    real(8), dimension(:,:), optional :: Hmat
    if (ed_total_ud) stop "ed_buildh_orbs ERROR: ed_total_ud=T takes ed_buildh_main"
    call buildh(isector, Hmat, "ed_buildh_orbs")
  end subroutine ed_buildh_orbs

  subroutine buildh(isector, Hmat, who)
    integer                           :: isector
    real(8), dimension(:,:), optional :: Hmat
    character(len=*)                  :: who
    real(8), allocatable, target      :: dense(:,:)
#ifdef _MPI
    ! ranks the shrunk communicator of a sector with DimDw < MpiSize leaves out (ED_HAMILTONIAN.f90:65-93) return at once,
    ! exactly as the reference's builders do (35-37, 218-220): no GPU sector, no collective
    if (MpiStatus .and. MpiComm == MPI_COMM_NULL) return
#endif
    if (.not. Hstatus) stop "ed_buildh ERROR: Hsector NOT set"
    isector = Hsector                                       ! as the reference does (40)
    if (present(Hmat)) then
       if (size(Hmat,1) /= getDim(isector) .or. size(Hmat,2) /= getDim(isector)) then
          write(*,"(A)") who//" ERROR: wrong shape of Hmat"
          stop "assert_shape"
       end if
    end if
    call gpu_build_live(merge(EDGPU_STORE_CSR, EDGPU_STORE_KRON, live_explicit_csr .and. ed_total_ud .and. DimPh == 1 .and. .not. Jhflag))
    if (present(Hmat)) then
       allocate(dense(size(Hmat,1), size(Hmat,2)))
       call edgpu_check(edgpu_sector_hmat(live_h, dense), who)
       Hmat = dense
    end if
  end subroutine buildh

  !> dd_sparse_HxV(Nloc,v,Hv): all four stored products are the one device operator (the handle knows the layout and the
  !  rank's dw columns); they differ, as in the reference (391-816), only in which build made the operator
  subroutine spMatVec_main(Nloc, v, Hv)
    integer                  :: Nloc
    real(8), dimension(Nloc) :: v, Hv
    call edgpu_live_hxv(Nloc, v, Hv)
  end subroutine spMatVec_main
  subroutine spMatVec_orbs(Nloc, v, Hv)
    integer                  :: Nloc
    real(8), dimension(Nloc) :: v, Hv
    call edgpu_live_hxv(Nloc, v, Hv)
  end subroutine spMatVec_orbs
  subroutine spMatVec_MPI_main(Nloc, v, Hv)
    integer                  :: Nloc
    real(8), dimension(Nloc) :: v, Hv
    iter = iter + 1                                        ! the reference counts its MPI products here (585)
    call edgpu_live_hxv(Nloc, v, Hv)
  end subroutine spMatVec_MPI_main
  subroutine spMatVec_MPI_orbs(Nloc, v, Hv)
    integer                  :: Nloc
    real(8), dimension(Nloc) :: v, Hv
    call edgpu_live_hxv(Nloc, v, Hv)
  end subroutine spMatVec_MPI_orbs

END MODULE ED_HAMILTONIAN_SPARSE_HxV
