!> TEST SCAFFOLDING ONLY -- not a build of the reference.  Minimal modules carrying the reference's module NAMES and just
!  the variables, types and helpers the same-name drop-in modules read (declarations after ED_INPUT_VARS.f90:13-78,
!  ED_VARS_GLOBAL.f90:11-160/257-284, ED_HAMILTONIAN_COMMON.f90:11-43; helpers restated from ED_SETUP.f90:446-500 and
!  ED_BATH/dmft_aux.f90:69-82), so that the drop-ins can be compile-checked and run against a host that looks like the
!  reference to them.  Nothing here is used by the product.
MODULE ED_INPUT_VARS
  implicit none
  integer              :: Norb = 1, Nspin = 1, Nbath = 5, Nph = 0
  real(8),dimension(5) :: Uloc = [2d0, 0d0, 0d0, 0d0, 0d0]
  real(8)              :: Ust = 0d0, Jh = 0d0, Jx = 0d0, Jp = 0d0, xmu = 0d0
  real(8),dimension(5) :: g_ph = 0d0
  real(8)              :: w0_ph = 0d0
  logical              :: HFmode = .true.
  logical              :: ed_sparse_H = .true.
  logical              :: ed_total_ud = .true.
  character(len=7)     :: bath_type = "normal"
END MODULE ED_INPUT_VARS

MODULE ED_VARS_GLOBAL
  USE ED_SPARSE_MATRIX
  USE MPI                                                   ! ED_VARS_GLOBAL.f90:4-7 (#ifdef _MPI)
  implicit none
  type H_operator
     real(8),dimension(:,:,:,:),allocatable :: O
  end type H_operator
  type(H_operator),dimension(:),allocatable :: H_basis
  type effective_bath_component
     integer                          :: N_dec
     real(8),dimension(:),allocatable :: v
     real(8),dimension(:),allocatable :: lambda
  end type effective_bath_component
  type effective_bath
     real(8),dimension(:,:,:),allocatable                    :: e
     real(8),dimension(:,:,:),allocatable                    :: v
     type(effective_bath_component),dimension(:),allocatable :: item
     logical                                                 :: status=.false.
  end type effective_bath
  type sector_map
     integer,dimension(:),allocatable :: map
     logical                          :: status=.false.
  end type sector_map
  abstract interface
     subroutine dd_sparse_HxV(Nloc,v,Hv)
       integer                 :: Nloc
       real(8),dimension(Nloc) :: v
       real(8),dimension(Nloc) :: Hv
     end subroutine dd_sparse_HxV
  end interface
  integer,save                                     :: Ns, Nsectors, Ns_orb, Ns_ud, DimPh
  real(8),dimension(:,:,:,:),allocatable           :: impHloc
  integer,allocatable,dimension(:)                 :: getDim
  type(effective_bath)                             :: dmft_bath
  type(sparse_matrix_csr)                          :: spH0d, spH0nd
  type(sparse_matrix_csr),dimension(:),allocatable :: spH0ups, spH0dws
  type(sparse_matrix_csr)                          :: spH0_ph, spH0e_eph, spH0ph_eph
  procedure(dd_sparse_HxV),pointer                 :: spHtimesV_p=>null()
  logical                                          :: Jhflag
  integer                                          :: MpiComm_Global = MPI_COMM_NULL    ! ED_VARS_GLOBAL.f90:265-266
  integer                                          :: MpiComm = MPI_COMM_NULL
  logical                                          :: MpiStatus = .false.
  integer                                          :: MpiRank = 0, MpiSize = 1
END MODULE ED_VARS_GLOBAL

MODULE ED_BATH
  USE ED_INPUT_VARS
  USE ED_VARS_GLOBAL
  implicit none
contains
  function bath_from_sym(lambdavec) result(Hbath)      ! sum of the symmetry-basis matrices weighted by lambda
    real(8),dimension(:)                      :: lambdavec
    real(8),dimension(Nspin,Nspin,Norb,Norb)  :: Hbath
    integer :: isym
    Hbath = 0d0
    do isym = 1, size(lambdavec)
       Hbath = Hbath + lambdavec(isym) * H_basis(isym)%O
    end do
  end function bath_from_sym
END MODULE ED_BATH

MODULE ED_SETUP
  USE ED_INPUT_VARS
  USE ED_VARS_GLOBAL
  implicit none
contains
  !> digits of isector-1 in base Ns_Orb+1, most significant first = [Nups, Ndws]
  subroutine sector_qn(isector, qn)
    integer :: isector, qn(2*Ns_Ud), count, k
    count = isector - 1
    do k = 2*Ns_Ud, 1, -1
       qn(k) = mod(count, Ns_Orb + 1); count = count / (Ns_Orb + 1)
    end do
  end subroutine sector_qn
  subroutine get_Nup(isector, Nup)
    integer :: isector, Nup(Ns_Ud), qn(2*Ns_Ud)
    call sector_qn(isector, qn); Nup = qn(1:Ns_Ud)
  end subroutine get_Nup
  subroutine get_Ndw(isector, Ndw)
    integer :: isector, Ndw(Ns_Ud), qn(2*Ns_Ud)
    call sector_qn(isector, qn); Ndw = qn(Ns_Ud+1:2*Ns_Ud)
  end subroutine get_Ndw
  integer function binom(n, k)
    integer :: n, k, i
    real(8) :: x
    x = 1d0
    do i = 1, k
       x = x * dble(n + 1 - i) / dble(i)
    end do
    binom = int(x + 0.5d0)
  end function binom
  subroutine setup_dimensions()
    integer :: isector, qn(10), k, d
    select case (trim(bath_type))
    case ("hybrid"); Ns = Nbath + Norb
    case default;    Ns = (Nbath + 1) * Norb
    end select
    if (ed_total_ud) then
       Ns_Orb = Ns; Ns_Ud = 1
    else
       Ns_Orb = Ns / Norb; Ns_Ud = Norb
    end if
    DimPh = Nph + 1
    Nsectors = ((Ns_Orb + 1)**2)**Ns_Ud
    Jhflag = Norb > 1 .and. (Jx /= 0d0 .or. Jp /= 0d0)
    if (allocated(getDim)) deallocate(getDim)
    allocate(getDim(Nsectors))
    do isector = 1, Nsectors
       call sector_qn(isector, qn(1:2*Ns_Ud))
       d = DimPh
       do k = 1, 2*Ns_Ud
          d = d * binom(Ns_Orb, qn(k))
       end do
       getDim(isector) = d
    end do
    if (allocated(spH0ups)) deallocate(spH0ups, spH0dws)
    allocate(spH0ups(Ns_Ud), spH0dws(Ns_Ud))
  end subroutine setup_dimensions
END MODULE ED_SETUP

MODULE ED_HAMILTONIAN_COMMON
  USE ED_INPUT_VARS
  USE ED_VARS_GLOBAL
  USE ED_BATH
  USE ED_SETUP
  implicit none
  integer                                   :: Dim
  integer                                   :: DimUp
  integer                                   :: DimDw
  integer,allocatable,dimension(:)          :: DimUps
  integer,allocatable,dimension(:)          :: DimDws
  integer                                   :: Hsector=0
  logical                                   :: Hstatus=.false.
  type(sector_map),dimension(:),allocatable :: Hs
  integer                                   :: Nfoo
  real(8),dimension(:,:,:),allocatable      :: diag_hybr
  real(8),dimension(:,:,:),allocatable      :: bath_diag
  integer,save,public                       :: iter=0
END MODULE ED_HAMILTONIAN_COMMON
