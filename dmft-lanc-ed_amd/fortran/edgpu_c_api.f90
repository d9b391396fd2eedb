!> ISO_C_BINDING view of include/edgpu.h -- the only place the Fortran host sees the C ABI.
!  Field order and kinds mirror `edgpu_desc` / `edgpu_info` exactly.
module EDGPU_C_API
  use, intrinsic :: iso_c_binding
  implicit none
  public

  integer(c_int32_t), parameter :: EDGPU_ABI_VERSION = 1
  integer(c_int32_t), parameter :: EDGPU_MAX_ORB = 5
  integer(c_int32_t), parameter :: EDGPU_BATH_NORMAL = 0, EDGPU_BATH_HYBRID = 1, EDGPU_BATH_REPLICA = 2
  integer(c_int32_t), parameter :: EDGPU_STORE_KRON = 0, EDGPU_STORE_CSR = 1, EDGPU_STORE_DIRECT = 2

  type, bind(C) :: edgpu_desc
     integer(c_int32_t) :: abi_version
     integer(c_int32_t) :: Norb, Nspin, Nbath
     integer(c_int32_t) :: Nph
     integer(c_int32_t) :: bath_type
     integer(c_int32_t) :: ed_total_ud
     integer(c_int32_t) :: storage
     integer(c_int32_t) :: hfmode
     integer(c_int32_t) :: Nfoo
     integer(c_int32_t) :: nups(EDGPU_MAX_ORB)
     integer(c_int32_t) :: ndws(EDGPU_MAX_ORB)
     integer(c_int32_t) :: mpi_rank, mpi_size
     integer(c_int32_t) :: device
     integer(c_int32_t) :: reserved
     real(c_double)     :: Uloc(EDGPU_MAX_ORB), Ust, Jh, Jx, Jp, xmu
     real(c_double)     :: w0_ph, g_ph(EDGPU_MAX_ORB)
     type(c_ptr)        :: impHloc, diag_hybr, bath_diag, hbath
  end type edgpu_desc

  type, bind(C) :: edgpu_info
     integer(c_int32_t) :: isector
     integer(c_int32_t) :: Ns, Ns_Orb, Ns_Ud, DimPh
     integer(c_int32_t) :: Jhflag
     integer(c_int32_t) :: DimUps(EDGPU_MAX_ORB), DimDws(EDGPU_MAX_ORB)
     integer(c_int64_t) :: DimUp, DimDw
     integer(c_int64_t) :: Dim
     integer(c_int64_t) :: mpiQdw
     integer(c_int64_t) :: mpiIshift
     integer(c_int64_t) :: vecDim
     integer(c_int64_t) :: nnz_total
     integer(c_int64_t) :: bytes_stored
  end type edgpu_info

  interface
     function edgpu_sector_create(desc, handle) bind(C, name="edgpu_sector_create") result(rc)
       import :: c_int, c_ptr, edgpu_desc
       type(edgpu_desc), intent(in) :: desc
       type(c_ptr), intent(out)     :: handle
       integer(c_int)               :: rc
     end function edgpu_sector_create
     function edgpu_sector_destroy(handle) bind(C, name="edgpu_sector_destroy") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: handle
       integer(c_int)     :: rc
     end function edgpu_sector_destroy
     function edgpu_sector_vecdim(handle) bind(C, name="edgpu_sector_vecdim") result(n)
       import :: c_int64_t, c_ptr
       type(c_ptr), value :: handle
       integer(c_int64_t) :: n
     end function edgpu_sector_vecdim
     function edgpu_sector_info(handle, info) bind(C, name="edgpu_sector_info") result(rc)
       import :: c_int, c_ptr, edgpu_info
       type(c_ptr), value            :: handle
       type(edgpu_info), intent(out) :: info
       integer(c_int)                :: rc
     end function edgpu_sector_info
     function edgpu_sector_map(handle, k, out) bind(C, name="edgpu_sector_map") result(rc)
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value        :: handle
       integer(c_int32_t), value :: k
       integer(c_int32_t)        :: out(*)
       integer(c_int)            :: rc
     end function edgpu_sector_map
     !> dd_sparse_HxV(Nloc,v,Hv), ED_VARS_GLOBAL.f90:75-81 : Nloc by reference, host arrays
     function edgpu_hxv(handle, nloc, v, hv) bind(C, name="edgpu_hxv") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value             :: handle
       integer(c_int32_t), intent(in) :: nloc
       real(c_double), intent(in)     :: v(*)
       real(c_double)                 :: hv(*)
       integer(c_int)                 :: rc
     end function edgpu_hxv
     function edgpu_lanczos_gs(handle, nitermax, threshold, ncheck, egs, vect, nlanc) bind(C, name="edgpu_lanczos_gs") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value        :: handle
       integer(c_int32_t), value :: nitermax, ncheck
       real(c_double), value     :: threshold
       real(c_double)            :: egs
       real(c_double)            :: vect(*)
       integer(c_int32_t)        :: nlanc
       integer(c_int)            :: rc
     end function edgpu_lanczos_gs
     function edgpu_lanczos_tridiag(handle, vin, nlanc, threshold, alanc, blanc, nused) bind(C, name="edgpu_lanczos_tridiag") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value         :: handle
       real(c_double), intent(in) :: vin(*)
       integer(c_int32_t), value  :: nlanc
       real(c_double), value      :: threshold
       real(c_double)             :: alanc(*), blanc(*)
       integer(c_int32_t)         :: nused
       integer(c_int)             :: rc
     end function edgpu_lanczos_tridiag
     function edgpu_eigsh(handle, nev, ncv, maxiter, tol, evals, evecs, nconv, nrestarts, nmatvec) bind(C, name="edgpu_eigsh") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value        :: handle
       integer(c_int32_t), value :: nev, ncv, maxiter
       real(c_double), value     :: tol
       real(c_double)            :: evals(*)
       real(c_double)            :: evecs(*)
       integer(c_int32_t)        :: nconv, nrestarts, nmatvec
       integer(c_int)            :: rc
     end function edgpu_eigsh
     function edgpu_dvdson(handle, nev, maxiter, tol, evals, evecs, nconv, nmatvec) bind(C, name="edgpu_dvdson") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value        :: handle
       integer(c_int32_t), value :: nev, maxiter
       real(c_double), value     :: tol
       real(c_double)            :: evals(*)
       real(c_double)            :: evecs(*)
       integer(c_int32_t)        :: nconv, nmatvec
       integer(c_int)            :: rc
     end function edgpu_dvdson
     function edgpu_gf_lanczos(hfrom, hto, iorb, ispin, dagger, state, nlanc, threshold, norm2, alanc, blanc, nused) &
          bind(C, name="edgpu_gf_lanczos") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value         :: hfrom, hto
       integer(c_int32_t), value  :: iorb, ispin, dagger, nlanc
       real(c_double), intent(in) :: state(*)
       real(c_double), value      :: threshold
       real(c_double)             :: norm2, alanc(*), blanc(*)
       integer(c_int32_t)         :: nused
       integer(c_int)             :: rc
     end function edgpu_gf_lanczos
     function edgpu_chi_lanczos(handle, cup, cdw, state, nlanc, threshold, norm2, alanc, blanc, nused) &
          bind(C, name="edgpu_chi_lanczos") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value         :: handle
       real(c_double), intent(in) :: cup(*), cdw(*), state(*)
       integer(c_int32_t), value  :: nlanc
       real(c_double), value      :: threshold
       real(c_double)             :: norm2, alanc(*), blanc(*)
       integer(c_int32_t)         :: nused
       integer(c_int)             :: rc
     end function edgpu_chi_lanczos
     function edgpu_phonon_lanczos(handle, state, nlanc, threshold, norm2, alanc, blanc, nused) &
          bind(C, name="edgpu_phonon_lanczos") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value         :: handle
       real(c_double), intent(in) :: state(*)
       integer(c_int32_t), value  :: nlanc
       real(c_double), value      :: threshold
       real(c_double)             :: norm2, alanc(*), blanc(*)
       integer(c_int32_t)         :: nused
       integer(c_int)             :: rc
     end function edgpu_phonon_lanczos
     function edgpu_imp_histogram(handle, vec, vec_on_device, hist) bind(C, name="edgpu_imp_histogram") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value         :: handle
       real(c_double), intent(in) :: vec(*)
       integer(c_int32_t), value  :: vec_on_device
       real(c_double)             :: hist(*)
       integer(c_int)             :: rc
     end function edgpu_imp_histogram
     function edgpu_imp_density_offdiag(handle, vec, dm) bind(C, name="edgpu_imp_density_offdiag") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value         :: handle
       real(c_double), intent(in) :: vec(*)
       real(c_double)             :: dm(*)
       integer(c_int)             :: rc
     end function edgpu_imp_density_offdiag
     function edgpu_local_energy(handle, state, out7) bind(C, name="edgpu_local_energy") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value         :: handle
       real(c_double), intent(in) :: state(*)
       real(c_double)             :: out7(7)
       integer(c_int)             :: rc
     end function edgpu_local_energy
     function edgpu_twin_reorder(handle, vin, vout) bind(C, name="edgpu_twin_reorder") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value         :: handle
       real(c_double), intent(in) :: vin(*)
       real(c_double)             :: vout(*)
       integer(c_int)             :: rc
     end function edgpu_twin_reorder
     function edgpu_comm_unique_id(id128) bind(C, name="edgpu_comm_unique_id") result(rc)
       import :: c_int, c_char
       character(kind=c_char) :: id128(128)
       integer(c_int)         :: rc
     end function edgpu_comm_unique_id
     function edgpu_comm_init_rccl(handle, id128) bind(C, name="edgpu_comm_init_rccl") result(rc)
       import :: c_int, c_char, c_ptr
       type(c_ptr), value                 :: handle
       character(kind=c_char), intent(in) :: id128(128)
       integer(c_int)                     :: rc
     end function edgpu_comm_init_rccl
     function edgpu_comm_init_callbacks(handle, allgatherv, exchange, allreduce, ctx) bind(C, name="edgpu_comm_init_callbacks") result(rc)
       import :: c_int, c_ptr, c_funptr
       type(c_ptr), value    :: handle
       type(c_funptr), value :: allgatherv, exchange, allreduce    ! bind(C) procedures, see include/edgpu.h
       type(c_ptr), value    :: ctx
       integer(c_int)        :: rc
     end function edgpu_comm_init_callbacks
     !> dense Hmat of build_Hv_sector(isector,Hmat): ED_HAMILTONIAN.f90:128-137
     function edgpu_sector_hmat(handle, hmat) bind(C, name="edgpu_sector_hmat") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: handle
       real(c_double)     :: hmat(*)
       integer(c_int)     :: rc
     end function edgpu_sector_hmat
     !> the process-wide communicator (MpiComm of ed_set_MpiComm, ED_VARS_GLOBAL.f90:332-347)
     function edgpu_comm_create_rccl(id128, rank, nranks, device, comm) bind(C, name="edgpu_comm_create_rccl") result(rc)
       import :: c_int, c_int32_t, c_char, c_ptr
       character(kind=c_char), intent(in) :: id128(128)
       integer(c_int32_t), value          :: rank, nranks, device
       type(c_ptr), intent(out)           :: comm
       integer(c_int)                     :: rc
     end function edgpu_comm_create_rccl
     function edgpu_comm_create_callbacks(rank, nranks, allgatherv, exchange, allreduce, ctx, comm) bind(C, name="edgpu_comm_create_callbacks") result(rc)
       import :: c_int, c_int32_t, c_ptr, c_funptr
       integer(c_int32_t), value :: rank, nranks
       type(c_funptr), value     :: allgatherv, exchange, allreduce
       type(c_ptr), value        :: ctx
       type(c_ptr), intent(out)  :: comm
       integer(c_int)            :: rc
     end function edgpu_comm_create_callbacks
     function edgpu_comm_sub(parent, nranks, comm, is_member) bind(C, name="edgpu_comm_sub") result(rc)
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value        :: parent
       integer(c_int32_t), value :: nranks
       type(c_ptr), intent(out)  :: comm
       integer(c_int32_t)        :: is_member
       integer(c_int)            :: rc
     end function edgpu_comm_sub
     function edgpu_comm_destroy(comm) bind(C, name="edgpu_comm_destroy") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: comm
       integer(c_int)     :: rc
     end function edgpu_comm_destroy
     function edgpu_sector_attach_comm(handle, comm) bind(C, name="edgpu_sector_attach_comm") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: handle, comm
       integer(c_int)     :: rc
     end function edgpu_sector_attach_comm
     function edgpu_comm_allreduce_sum(comm, buf, n) bind(C, name="edgpu_comm_allreduce_sum") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value        :: comm
       real(c_double)            :: buf(*)
       integer(c_int32_t), value :: n
       integer(c_int)            :: rc
     end function edgpu_comm_allreduce_sum
     !> sparse container (ED_SPARSE_MATRIX.f90): see include/edgpu.h
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
       import :: c_int, c_int32_t, c_int64_t, c_ptr
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
     function edgpu_sp_matvec_device(m, v_dev, hv_dev, accumulate, hip_stream) bind(C, name="edgpu_sp_matvec_device") result(rc)
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value         :: m, v_dev, hv_dev, hip_stream     ! device pointers; hip_stream: c_null_ptr = default stream
       integer(c_int32_t), value  :: accumulate
       integer(c_int)             :: rc
     end function edgpu_sp_matvec_device
     function edgpu_cache_release() bind(C, name="edgpu_cache_release") result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function edgpu_cache_release
     function edgpu_last_error() bind(C, name="edgpu_last_error") result(msg)
       import :: c_ptr
       type(c_ptr) :: msg
     end function edgpu_last_error
     function edgpu_device_count(n) bind(C, name="edgpu_device_count") result(rc)
       import :: c_int, c_int32_t
       integer(c_int32_t) :: n
       integer(c_int)     :: rc
     end function edgpu_device_count
     ! ---- device-resident flow: vectors stay in HBM between calls (type(c_ptr) = device address) ----------------------
     function edgpu_malloc(dev, bytes) bind(C, name="edgpu_malloc") result(rc)
       import :: c_int, c_size_t, c_ptr
       type(c_ptr)              :: dev
       integer(c_size_t), value :: bytes
       integer(c_int)           :: rc
     end function edgpu_malloc
     function edgpu_free(dev) bind(C, name="edgpu_free") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: dev
       integer(c_int)     :: rc
     end function edgpu_free
     function edgpu_memcpy_h2d(dev, host, bytes) bind(C, name="edgpu_memcpy_h2d") result(rc)
       import :: c_int, c_size_t, c_ptr, c_double
       type(c_ptr), value         :: dev
       real(c_double), intent(in) :: host(*)
       integer(c_size_t), value   :: bytes
       integer(c_int)             :: rc
     end function edgpu_memcpy_h2d
     function edgpu_memcpy_d2h(host, dev, bytes) bind(C, name="edgpu_memcpy_d2h") result(rc)
       import :: c_int, c_size_t, c_ptr, c_double
       real(c_double)           :: host(*)
       type(c_ptr), value       :: dev
       integer(c_size_t), value :: bytes
       integer(c_int)           :: rc
     end function edgpu_memcpy_d2h
     function edgpu_memset(dev, byte_value, bytes) bind(C, name="edgpu_memset") result(rc)
       import :: c_int, c_int32_t, c_size_t, c_ptr
       type(c_ptr), value        :: dev
       integer(c_int32_t), value :: byte_value
       integer(c_size_t), value  :: bytes
       integer(c_int)            :: rc
     end function edgpu_memset
     function edgpu_device_synchronize() bind(C, name="edgpu_device_synchronize") result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function edgpu_device_synchronize
     function edgpu_hxv_device(handle, v_dev, hv_dev) bind(C, name="edgpu_hxv_device") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: handle, v_dev, hv_dev
       integer(c_int)     :: rc
     end function edgpu_hxv_device
     function edgpu_lanczos_gs_device(handle, nitermax, threshold, ncheck, egs, vect_dev, nlanc) &
          bind(C, name="edgpu_lanczos_gs_device") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value        :: handle, vect_dev
       integer(c_int32_t), value :: nitermax, ncheck
       real(c_double), value     :: threshold
       real(c_double)            :: egs
       integer(c_int32_t)        :: nlanc
       integer(c_int)            :: rc
     end function edgpu_lanczos_gs_device
     function edgpu_lanczos_tridiag_device(handle, vin_dev, nlanc, threshold, alanc, blanc, nused) &
          bind(C, name="edgpu_lanczos_tridiag_device") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value        :: handle, vin_dev
       integer(c_int32_t), value :: nlanc
       real(c_double), value     :: threshold
       real(c_double)            :: alanc(*), blanc(*)
       integer(c_int32_t)        :: nused
       integer(c_int)            :: rc
     end function edgpu_lanczos_tridiag_device
     function edgpu_apply_op(hfrom, hto, iorb, ispin, dagger, vin_dev, vout_dev, norm2) bind(C, name="edgpu_apply_op") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value        :: hfrom, hto, vin_dev, vout_dev
       integer(c_int32_t), value :: iorb, ispin, dagger
       real(c_double)            :: norm2
       integer(c_int)            :: rc
     end function edgpu_apply_op
     ! ---- introspection --------------------------------------------------------------------------------------------
     function edgpu_abi_version_of_library() bind(C, name="edgpu_abi_version") result(v)      ! compare with the parameter EDGPU_ABI_VERSION
       import :: c_int32_t
       integer(c_int32_t) :: v
     end function edgpu_abi_version_of_library
     function edgpu_eigsh_ncv_cap() bind(C, name="edgpu_eigsh_ncv_cap") result(n)
       import :: c_int32_t
       integer(c_int32_t) :: n
     end function edgpu_eigsh_ncv_cap
     function edgpu_eigsh_ncv_used(handle) bind(C, name="edgpu_eigsh_ncv_used") result(n)
       import :: c_int32_t, c_ptr
       type(c_ptr), value :: handle
       integer(c_int32_t) :: n
     end function edgpu_eigsh_ncv_used
     function edgpu_lanczos_is_fused(handle) bind(C, name="edgpu_lanczos_is_fused") result(n)
       import :: c_int32_t, c_ptr
       type(c_ptr), value :: handle
       integer(c_int32_t) :: n
     end function edgpu_lanczos_is_fused
     function edgpu_sector_describe(handle, buf, cap) bind(C, name="edgpu_sector_describe") result(rc)
       import :: c_int, c_int32_t, c_char, c_ptr
       type(c_ptr), value        :: handle
       character(kind=c_char)    :: buf(*)
       integer(c_int32_t), value :: cap
       integer(c_int)            :: rc
     end function edgpu_sector_describe
     function edgpu_comm_count(n) bind(C, name="edgpu_comm_count") result(rc)
       import :: c_int, c_int32_t
       integer(c_int32_t) :: n
       integer(c_int)     :: rc
     end function edgpu_comm_count
     function edgpu_sector_exchange_info(handle, mode, send_bytes, recv_bytes) bind(C, name="edgpu_sector_exchange_info") result(rc)
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       type(c_ptr), value :: handle
       integer(c_int32_t) :: mode
       integer(c_int64_t) :: send_bytes(*), recv_bytes(*)      ! one entry per rank
       integer(c_int)     :: rc
     end function edgpu_sector_exchange_info
     function edgpu_sector_relay_info(handle, enabled, nchunks, busiest_before, busiest_after, rows_direct, rows_routed) &
          bind(C, name="edgpu_sector_relay_info") result(rc)
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       type(c_ptr), value :: handle
       integer(c_int32_t) :: enabled, nchunks
       integer(c_int64_t) :: busiest_before, busiest_after
       integer(c_int64_t) :: rows_direct(*), rows_routed(*)    ! MpiSize*MpiSize entries, [from][to] row-major
       integer(c_int)     :: rc
     end function edgpu_sector_relay_info
  end interface

contains

  !> the reference aborts with `stop "<msg>"`: non-zero status -> print the library's message and stop
  subroutine edgpu_check(rc, where)
    integer(c_int), intent(in)   :: rc
    character(len=*), intent(in) :: where
    if (rc /= 0) then
       write(*,"(A)") trim(where)//": "//edgpu_error_message()
       stop "edgpu error"
    end if
  end subroutine edgpu_check

  !> message of the last failed call as a Fortran string
  function edgpu_error_message() result(s)
    character(len=:), allocatable   :: s
    type(c_ptr)                     :: p
    character(kind=c_char), pointer :: cs(:)
    integer                         :: n
    p = edgpu_last_error()
    s = ""
    if (.not. c_associated(p)) return
    call c_f_pointer(p, cs, [1024])
    n = 0
    do while (n < 1024)
       if (cs(n+1) == c_null_char) exit
       n = n + 1
    end do
    allocate(character(len=n) :: s)
    block
      integer :: i
      do i = 1, n
         s(i:i) = cs(i)
      end do
    end block
  end function edgpu_error_message

end module EDGPU_C_API
