!> Stand-in for the four SciFortran (SF_SP_LINALG) entry points the reference calls with its sector operator --
!     sp_eigh          ED_DIAG.f90:153-170      (ARPACK, the default lanc_method)
!     sp_lanc_eigh     ED_DIAG.f90:177-185
!     sp_dvdson_eigh   ED_DIAG.f90:189-203
!     sp_lanc_tridiag  ED_GF_NORMAL.f90:232-237, ED_GF_PHONON.f90:110-115, ED_GF_CHISPIN/CHIDENS.f90
!  with the reference's call shapes (MatVec first, or MpiComm then MatVec), running the WHOLE loop on the live GPU sector:
!  vectors stay in HBM, only eigenvalues, Lanczos coefficients and the final vectors cross PCIe.  Put this module ahead of
!  SciFortran's on the include path and ED_DIAG / ED_GF_* compile unchanged.  `MatVec` must be the operator of the live
!  sector (the reference never passes anything else: spHtimesV_p); with ed_sparse_H=F the sector does not exist until the
!  first product, so MatVec is called once, on the start vector, to bring it to life.
module SF_SP_LINALG
  use, intrinsic :: iso_c_binding
#ifdef _MPI
  use MPI, only: MPI_COMM_NULL
#endif
  use EDGPU_C_API
  use EDGPU_LIVE
  implicit none
  private

  abstract interface
     subroutine sf_matvec(Nloc, v, Hv)
       integer                  :: Nloc
       real(8), dimension(Nloc) :: v, Hv
     end subroutine sf_matvec
  end interface

  interface sp_eigh
     module procedure :: sp_eigh_serial, sp_eigh_mpi
  end interface
  interface sp_lanc_eigh
     module procedure :: sp_lanc_eigh_serial, sp_lanc_eigh_mpi
  end interface
  interface sp_dvdson_eigh
     module procedure :: sp_dvdson_eigh_serial
  end interface
  interface sp_lanc_tridiag
     module procedure :: sp_lanc_tridiag_serial, sp_lanc_tridiag_mpi
  end interface

  public :: sp_eigh, sp_lanc_eigh, sp_dvdson_eigh, sp_lanc_tridiag

contains

  !> SciFortran's MPI forms return at once on a rank whose communicator is MPI_COMM_NULL -- the ranks a sector with
  !  DimDw < MpiSize leaves out call them all the same (ED_DIAG.f90:153-185 runs on every rank)
  logical function left_out(MpiComm)
    integer, intent(in) :: MpiComm
    left_out = .false.
#ifdef _MPI
    left_out = MpiComm == MPI_COMM_NULL
#endif
  end function left_out

  subroutine wake(MatVec, n, v)
    procedure(sf_matvec)   :: MatVec
    integer, intent(in)    :: n
    real(8), intent(inout) :: v(n)
    real(8), allocatable   :: tmp(:)
    if (live_status) return
    allocate(tmp(n))
    call MatVec(n, v, tmp)                  ! directMatVec_*: builds the on-the-fly GPU sector
    call edgpu_live_require("SF_SP_LINALG")
  end subroutine wake

  subroutine sp_eigh_serial(MatVec, eval, evec, Nblock, Nitermax, v0, tol, iverbose)
    procedure(sf_matvec)          :: MatVec
    real(8)                       :: eval(:)
    real(8)                       :: evec(:,:)        ! (vecDim, Neigen)
    integer                       :: Nblock, Nitermax
    real(8), optional             :: v0(size(evec,1))
    real(8), optional             :: tol
    logical, optional             :: iverbose
    real(c_double)     :: tl
    integer(c_int32_t) :: nc, nr, nm
    real(8), allocatable, target :: ev(:,:)
    tl = 0d0; if (present(tol)) tl = tol
    call wake(MatVec, size(evec,1), evec(:,1))
    if (size(evec,1) /= live_info%vecDim .or. size(evec,2) /= size(eval)) stop "sp_eigh ERROR: wrong shape of evec"
    allocate(ev(size(evec,1), size(evec,2)))
    call edgpu_check(edgpu_eigsh(live_h, int(size(eval), c_int32_t), int(Nblock, c_int32_t), int(Nitermax, c_int32_t), tl, &
         eval, ev, nc, nr, nm), "sp_eigh")
    evec = ev
    if (present(iverbose)) then
       if (iverbose) write(*,"(A,I6,A,I6,A,I8)") "sp_eigh (GPU thick-restart Lanczos): converged", nc, " restarts", nr, " products", nm
    end if
  end subroutine sp_eigh_serial
  subroutine sp_eigh_mpi(MpiComm, MatVec, eval, evec, Nblock, Nitermax, v0, tol, iverbose)
    integer                       :: MpiComm
    procedure(sf_matvec)          :: MatVec
    real(8)                       :: eval(:)
    real(8)                       :: evec(:,:)
    integer                       :: Nblock, Nitermax
    real(8), optional             :: v0(size(evec,1))
    real(8), optional             :: tol
    logical, optional             :: iverbose
    if (left_out(MpiComm)) return
    call sp_eigh_serial(MatVec, eval, evec, Nblock, Nitermax, v0, tol, iverbose)   ! the sector carries its communicator
  end subroutine sp_eigh_mpi

  subroutine sp_lanc_eigh_serial(MatVec, egs, vect, Nitermax, iverbose, threshold, ncheck)
    procedure(sf_matvec)          :: MatVec
    real(8)                       :: egs
    real(8)                       :: vect(:)
    integer                       :: Nitermax
    logical, optional             :: iverbose
    real(8), optional             :: threshold
    integer, optional             :: ncheck
    real(c_double)     :: thr
    integer(c_int32_t) :: nchk, nl
    real(8), allocatable :: vv(:)
    thr = 1d-12; if (present(threshold)) thr = threshold
    nchk = 10;   if (present(ncheck)) nchk = ncheck
    vv = vect
    call wake(MatVec, size(vv), vv)
    if (size(vect) /= live_info%vecDim) stop "sp_lanc_eigh ERROR: size(vect) != vecDim"
    call edgpu_check(edgpu_lanczos_gs(live_h, int(Nitermax, c_int32_t), thr, nchk, egs, vv, nl), "sp_lanc_eigh")
    vect = vv
    if (present(iverbose)) then
       if (iverbose) write(*,"(A,I6,A,F20.12)") "sp_lanc_eigh (GPU): iterations", nl, " E0", egs
    end if
  end subroutine sp_lanc_eigh_serial
  subroutine sp_lanc_eigh_mpi(MpiComm, MatVec, egs, vect, Nitermax, iverbose, threshold, ncheck)
    integer                       :: MpiComm
    procedure(sf_matvec)          :: MatVec
    real(8)                       :: egs
    real(8)                       :: vect(:)
    integer                       :: Nitermax
    logical, optional             :: iverbose
    real(8), optional             :: threshold
    integer, optional             :: ncheck
    if (left_out(MpiComm)) return
    call sp_lanc_eigh_serial(MatVec, egs, vect, Nitermax, iverbose, threshold, ncheck)
  end subroutine sp_lanc_eigh_mpi

  subroutine sp_dvdson_eigh_serial(MatVec, eval, evec, Nblock, Nitermax, Tol)
    procedure(sf_matvec)          :: MatVec
    real(8)                       :: eval(:)
    real(8)                       :: evec(:,:)
    integer, optional             :: Nblock, Nitermax
    real(8), optional             :: Tol
    real(c_double)     :: tl
    integer(c_int32_t) :: nc, nm, nit
    real(8), allocatable, target :: ev(:,:)
    tl = 0d0; if (present(Tol)) tl = Tol
    nit = 512; if (present(Nitermax)) nit = int(Nitermax, c_int32_t)
    call wake(MatVec, size(evec,1), evec(:,1))
    if (size(evec,1) /= live_info%vecDim .or. size(evec,2) /= size(eval)) stop "sp_dvdson_eigh ERROR: wrong shape of evec"
    allocate(ev(size(evec,1), size(evec,2)))
    call edgpu_check(edgpu_dvdson(live_h, int(size(eval), c_int32_t), nit, tl, eval, ev, nc, nm), "sp_dvdson_eigh")
    evec = ev
  end subroutine sp_dvdson_eigh_serial

  subroutine sp_lanc_tridiag_serial(MatVec, vin, alanc, blanc, threshold)
    procedure(sf_matvec)          :: MatVec
    real(8)                       :: vin(:)
    real(8)                       :: alanc(:), blanc(:)
    real(8), optional             :: threshold
    real(c_double)     :: thr
    integer(c_int32_t) :: nu
    thr = 1d-12; if (present(threshold)) thr = threshold
    if (size(alanc) /= size(blanc)) stop "sp_lanc_tridiag ERROR: size(alanc) != size(blanc)"
    call wake(MatVec, size(vin), vin)
    if (size(vin) /= live_info%vecDim) stop "sp_lanc_tridiag ERROR: size(vin) != vecDim"
    call edgpu_check(edgpu_lanczos_tridiag(live_h, vin, int(size(alanc), c_int32_t), thr, alanc, blanc, nu), "sp_lanc_tridiag")
  end subroutine sp_lanc_tridiag_serial
  subroutine sp_lanc_tridiag_mpi(MpiComm, MatVec, vin, alanc, blanc, threshold)
    integer                       :: MpiComm
    procedure(sf_matvec)          :: MatVec
    real(8)                       :: vin(:)
    real(8)                       :: alanc(:), blanc(:)
    real(8), optional             :: threshold
    if (left_out(MpiComm)) return
    call sp_lanc_tridiag_serial(MatVec, vin, alanc, blanc, threshold)
  end subroutine sp_lanc_tridiag_mpi

end module SF_SP_LINALG
