/*
 * edgpu.h -- C ABI of libedgpu.so, the MI355X (gfx950) implementation of the
 * Lanczos-ED hot path of lcrippa/dmft-lanc-ed.
 *
 * Every entry point names the reference interface it replaces (file:line in
 * /root/reference).  Plain pointers and sizes only; arrays are column-major
 * with the reference's index order so the unchanged Fortran host can pass
 * them as they are (see INTEGRATION.md for the ISO_C_BINDING stubs).
 *
 * Error convention: the reference aborts with `stop "<msg>"`; here every
 * function returns 0 on success or a negative EDGPU_E* code, and
 * edgpu_last_error() returns the message the Fortran shim prints before
 * `stop`.  There is no CPU fallback: without a HIP device calls that compute
 * fail with EDGPU_ENODEVICE.
 *
 * Threading: one host thread per rank/GPU, one sector live per handle
 * (the reference keeps one sector in module globals, ED_HAMILTONIAN_COMMON.f90:11-43).
 */
#ifndef EDGPU_H
#define EDGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EDGPU_ABI_VERSION 1
#define EDGPU_MAX_ORB 5 /* ED_SETUP.f90:66 "Norb > 5 is currently not supported" */

enum { EDGPU_OK = 0, EDGPU_EINVAL = -1, EDGPU_ENODEVICE = -2, EDGPU_EHIP = -3, EDGPU_ENOMEM = -4,
       EDGPU_EUNSUPPORTED = -5, EDGPU_ECOMM = -6, EDGPU_ENOTCONVERGED = -7 };

/* bath_type, ED_INPUT_VARS.f90:205 */
enum { EDGPU_BATH_NORMAL = 0, EDGPU_BATH_HYBRID = 1, EDGPU_BATH_REPLICA = 2 };

/* How the sector operator is held in HBM.
 *  KRON   : the reference's stored format (ed_sparse_H=T): diagonal spH0d + per-spin
 *           factors spH0ups/spH0dws (+ spH0nd, phonon blocks); ED_VARS_GLOBAL.f90:142-146.
 *  CSR    : the same operator expanded to explicit CSR (int64 row_ptr, int32 col, fp64 val): for large sectors as a local
 *           block + a dw block; Jx/Jp sectors add the stored non-local block (spH0nd) as compressed rows, phonon sectors
 *           the stored electronic factor of the e-ph coupling (spH0e_eph); ed_total_ud=F through composite up/dw blocks.
 *  DIRECT : no stored operator (ed_sparse_H=F): the O(N) diagonal is recomputed from the bit patterns inside every product; the
 *           O(DimUp + DimDw) hop tables of the spin factors are derived from the bit patterns by the first product of a sector
 *           (EDGPU_DIRECT_CACHE=0 in the environment: by every product); ED_HAMILTONIAN_DIRECT_HxV.f90:21-95. */
enum { EDGPU_STORE_KRON = 0, EDGPU_STORE_CSR = 1, EDGPU_STORE_DIRECT = 2 };

/* Resolved inputs of build_Hv_sector (ED_HAMILTONIAN.f90:43-168) -- the values the
 * Fortran side holds in ED_INPUT_VARS / ED_VARS_GLOBAL when it builds a sector. */
typedef struct edgpu_desc {
    int32_t abi_version;          /* EDGPU_ABI_VERSION */
    int32_t Norb, Nspin, Nbath;   /* ED_INPUT_VARS.f90:129-131 */
    int32_t Nph;                  /* DimPh = Nph+1, ED_SETUP.f90:133 */
    int32_t bath_type;            /* EDGPU_BATH_* */
    int32_t ed_total_ud;          /* logical, ED_INPUT_VARS.f90:152 */
    int32_t storage;              /* EDGPU_STORE_*; ed_sparse_H=T -> KRON (or CSR), F -> DIRECT */
    int32_t hfmode;               /* logical, ED_INPUT_VARS.f90:176 */
    int32_t Nfoo;                 /* size(bath_diag,2): Norb (normal,replica) or 1 (hybrid) */
    int32_t nups[EDGPU_MAX_ORB];  /* sector quantum numbers, Ns_Ud entries used */
    int32_t ndws[EDGPU_MAX_ORB];
    int32_t mpi_rank, mpi_size;   /* dw-column partition, ED_HAMILTONIAN.f90:96-110 */
    int32_t device;               /* HIP device ordinal, -1 = current device */
    int32_t reserved;             /* bit 0: run the exchange code path even with mpi_size=1 (self-test) */
    double Uloc[EDGPU_MAX_ORB], Ust, Jh, Jx, Jp, xmu;
    double w0_ph, g_ph[EDGPU_MAX_ORB];
    const double *impHloc;   /* (Nspin,Nspin,Norb,Norb)        ED_VARS_GLOBAL.f90:120 */
    const double *diag_hybr; /* (Nspin,Norb,Nbath)             ED_HAMILTONIAN_COMMON.f90:41 */
    const double *bath_diag; /* (Nspin,Nfoo,Nbath)             ED_HAMILTONIAN_COMMON.f90:42 */
    const double *hbath;     /* (Nspin,Nspin,Norb,Norb,Nbath)  replica only, else NULL */
} edgpu_desc;

typedef struct edgpu_sector_s *edgpu_handle;

typedef struct edgpu_info {
    int32_t isector;              /* ED_SETUP.f90:446-457 */
    int32_t Ns, Ns_Orb, Ns_Ud, DimPh;
    int32_t Jhflag;               /* ED_SETUP.f90:147-148 */
    int32_t DimUps[EDGPU_MAX_ORB], DimDws[EDGPU_MAX_ORB];
    int64_t DimUp, DimDw;         /* products */
    int64_t Dim;                  /* getDim(isector) = DimUp*DimDw*DimPh (64-bit: Ns=18 overflows int32) */
    int64_t mpiQdw;               /* local dw columns */
    int64_t mpiIshift;            /* global offset of the first local electronic row */
    int64_t vecDim;               /* DimUp*mpiQdw*DimPh, ED_HAMILTONIAN.f90:251 */
    int64_t nnz_total;            /* nnz of the expanded operator restricted to local rows */
    int64_t bytes_stored;         /* HBM bytes held by the operator in the chosen storage */
} edgpu_info;

/* ---- lifecycle: build_Hv_sector / delete_Hv_sector / vecDim_Hv_sector -------------
 * ED_HAMILTONIAN.f90:43-168, 174-222, 229-253 */
int edgpu_sector_create(const edgpu_desc *desc, edgpu_handle *out);
int edgpu_sector_destroy(edgpu_handle h);
int64_t edgpu_sector_vecdim(edgpu_handle h);
int edgpu_sector_info(edgpu_handle h, edgpu_info *out);

/* ---- bit-exact inspection of what build_sector / ed_buildh_* produced ---------------
 * Hs(k)%map, k=0..2*Ns_Ud-1 ordered [ups..., dws...]   (ED_SETUP.f90:745-780) */
int edgpu_sector_map(edgpu_handle h, int32_t k, int32_t *out);
/* spH0ups(iud)/spH0dws(iud)/spH0nd as (row_ptr[nrow+1], cols 1-based, vals), rows in the
 * reference's storage order (ED_SPARSE_MATRIX.f90:255-323).  which: 1 nd, 2 up, 3 dw. */
enum { EDGPU_SP_ND = 1, EDGPU_SP_UP = 2, EDGPU_SP_DW = 3 };
int edgpu_factor_size(edgpu_handle h, int32_t which, int32_t iud, int64_t *nrow, int64_t *nnz);
int edgpu_factor_export(edgpu_handle h, int32_t which, int32_t iud, int64_t *row_ptr, int64_t *cols, double *vals);
/* spH0d: the local diagonal d(1:DimUp*mpiQdw) (stored/H_local.f90:13-71), computed on the device */
int edgpu_diag_export(edgpu_handle h, double *d_host);
/* dense Hmat of build_Hv_sector(isector,Hmat) (ED_HAMILTONIAN.f90:128-137; ed_buildh_main/orbs with Hmat,
 * ED_HAMILTONIAN_SPARSE_HxV.f90:132-195, 304-363): the form the reference diagonalises with LAPACK for every sector
 * with dim <= lanc_dim_threshold (ED_DIAG.f90:105-107, 212-237).  hmat: host, Dim x Dim column-major, overwritten.
 * With more than one rank the rows are summed over the ranks (mpi_sp_dump_matrix, ED_SPARSE_MATRIX.f90:408-437):
 * every rank gets the full matrix. */
int edgpu_sector_hmat(edgpu_handle h, double *hmat);
/* explicit CSR (storage=CSR only): sizes and a window of rows [row0,row0+nrows) copied to the host as plain CSR rows --
 * ascending columns in the dense numbering jup + jdw*DimUp, whatever the device layout (blocks, lane-major 64-row blocks,
 * tile-major dw storage, pitched column space) */
int edgpu_csr_rows(edgpu_handle h, int64_t row0, int64_t nrows, int64_t *row_ptr, int32_t *cols, double *vals, int64_t cap);

/* ---- the operator: dd_sparse_HxV(Nloc,v,Hv), ED_VARS_GLOBAL.f90:75-81 ----------------
 * Host-pointer form, callable through spHtimesV_p.  Overwrites Hv.  Copies v in and Hv
 * out over PCIe on every call: correctness path, not the fast path. */
int edgpu_hxv(edgpu_handle h, const int32_t *nloc, const double *v, double *hv);
int edgpu_hxv64(edgpu_handle h, int64_t nloc, const double *v, double *hv);
/* Device-pointer form: v and Hv already live in HBM (vecDim doubles each). */
int edgpu_hxv_device(edgpu_handle h, const double *v_dev, double *hv_dev);
/* Mean device time (ms) of one product over `nrep` launches after `nwarm`, measured with
 * HIP events on the stream the kernels are launched on. */
int edgpu_hxv_time(edgpu_handle h, const double *v_dev, double *hv_dev, int32_t nwarm, int32_t nrep, float *ms_mean);

/* ---- the Lanczos loop the reference delegates to SciFortran ---------------------------
 * sp_lanc_eigh(MatVec,egs,vect,Nitermax,iverbose,threshold)   ED_DIAG.f90:177-185
 * vect (host, vecDim): in = start vector (all zero -> deterministic pseudo-random start,
 * as the library does), out = normalised ground state.  The whole loop runs in HBM. */
int edgpu_lanczos_gs(edgpu_handle h, int32_t nitermax, double threshold, int32_t ncheck,
                     double *egs, double *vect, int32_t *nlanc);
/* same with the vector resident on the device (no PCIe traffic) */
int edgpu_lanczos_gs_device(edgpu_handle h, int32_t nitermax, double threshold, int32_t ncheck,
                            double *egs, double *vect_dev, int32_t *nlanc);
/* sp_lanc_tridiag(MatVec,vin,alanc,blanc)   ED_GF_NORMAL.f90:232-237 ; blanc(1) unused (638-640) */
int edgpu_lanczos_tridiag(edgpu_handle h, const double *vin, int32_t nlanc, double threshold,
                          double *alanc, double *blanc, int32_t *nused);
int edgpu_lanczos_tridiag_device(edgpu_handle h, const double *vin_dev, int32_t nlanc, double threshold,
                                 double *alanc, double *blanc, int32_t *nused);
/* `niter` plain Lanczos iterations from a fixed start, no convergence test, no host sync
 * inside the loop: the timed region of bench.py.  Returns mean ms/iteration (HIP events). */
int edgpu_lanczos_time(edgpu_handle h, int32_t nwarm, int32_t niter, float *ms_per_iter);

/* sp_eigh(MatVec,eval,evec,Nblock,Nitermax,tol)   ED_DIAG.f90:153-170 (ARPACK dsaupd/dseupd, which='SA'):
 * the lowest `nev` eigenpairs from a thick-restart Lanczos with a basis of `ncv` (=Nblock) vectors kept in
 * HBM; `maxiter` restarts at most.  evals[nev] ascending; evecs (host, vecDim x nev column-major, may be
 * NULL).  nconv = number of requested pairs whose residual passed |r| <= tol*max(eps^(2/3),|theta|). */
int edgpu_eigsh(edgpu_handle h, int32_t nev, int32_t ncv, int32_t maxiter, double tol, double *evals, double *evecs,
                int32_t *nconv, int32_t *nrestarts, int32_t *nmatvec);
/* Returns EDGPU_ENOTCONVERGED (outputs still filled) when fewer than nev pairs converged.  The basis holds at most
 * edgpu_eigsh_ncv_cap() vectors (96; the reference's Nblock = 10*max(Neigen,lanc_nstates_sector)+lanc_ncv_add,
 * ED_DIAG.f90:95-98, exceeds that only for lanc_nstates_sector > 9): a larger ncv is reduced to the cap -- a smaller
 * thick-restart basis costs restarts, not accuracy -- and edgpu_eigsh_ncv_used() tells what the last call ran with. */
int32_t edgpu_eigsh_ncv_cap(void);
int32_t edgpu_eigsh_ncv_used(edgpu_handle h);
/* 1 when this sector's Lanczos loops (edgpu_lanczos_gs / _tridiag and the searches built on them) run the fused two-pass
 * iteration, 0 when they run product + axpy/norm as three passes (EDGPU_LZ_FUSED=0, multi-rank, CSR storage, non-local terms). */
int32_t edgpu_lanczos_is_fused(edgpu_handle h);
/* One-line description of the forms this sector's product and Lanczos loops take (storage, one- or two-pass product,
 * local-pass kernel, composite blocks, fused iteration, row pitch of the work vectors, exchange mode): written to buf
 * (NUL-terminated, truncated to cap).  For logs and bench records; no reference counterpart. */
int edgpu_sector_describe(edgpu_handle h, char *buf, int32_t cap);
/* sp_dvdson_eigh(MatVec,eval,evec,nitermax,tol)   ED_DIAG.f90:189-203 (lanc_method="dvdson"): Davidson with the
 * diagonal of H as preconditioner; search space (2*nev + 4 vectors, restarted with the previous iterate kept) and its
 * image under H in HBM.  `maxiter` products at most; a pair
 * counts as converged when |H x - theta x| <= max(tol,1e-11)*max(1,|theta|).  Same outputs as edgpu_eigsh. */
int edgpu_dvdson(edgpu_handle h, int32_t nev, int32_t maxiter, double tol, double *evals, double *evecs,
                 int32_t *nconv, int32_t *nmatvec);      /* EDGPU_ENOTCONVERGED as edgpu_eigsh */

/* ---- observables: impurity-configuration weights of a state --------------------
 * hist[(mu*2^Norb + md) + 4^Norb*iph] = sum_i |vec_i|^2 over the states whose impurity up/down
 * occupation words are mu/md (bit a-1 = orbital a) and phonon number iph.  dens, dens_up/dw, docc, magz,
 * sz2, n2, s2tot, Prob, prob_ph, dens_ph of lanc_observables (ED_OBSERVABLES.f90:150-215) are sums over
 * this table, so the N-vector stays in HBM.  vec: host (vec_on_device=0) or device pointer, vecDim doubles. */
int edgpu_imp_histogram(edgpu_handle h, const double *vec, int32_t vec_on_device, double *hist);

/* off-diagonal impurity density matrix <c^+_a c_b>_s of a state (ED_OBSERVABLES.f90:283-313, ed_total_ud=T only):
 * dm[(s*Norb + a)*Norb + b], s = 0 (up) / 1 (dw), a != b; the diagonal follows from edgpu_imp_histogram. */
int edgpu_imp_density_offdiag(edgpu_handle h, const double *vec_host, double *dm);

/* local energy terms of one state (lanc_local_energy, ED_OBSERVABLES.f90:372-616), weight 1 -- the host applies
 * peso = exp(-beta(Ei-Egs))/Z and sums over its state list: out7 = { ed_Epot (interaction energy including the Hartree shift,
 * as line 589 leaves it), ed_Ehartree, ed_Eknot, ed_Dust, ed_Dund, ed_Dse, ed_Dph }.  Density terms from the
 * impurity-configuration weights, the off-diagonal impHloc part of Eknot from the impurity density matrix, the spin-exchange
 * and pair-hopping expectation values (Dse, Dph; only when Jx / Jp are non-zero, as in the reference) from one pass over the
 * state in HBM.  state: host or device pointer, vecDim doubles; summed over ranks. */
int edgpu_local_energy(edgpu_handle h, const double *state_host, double *out7);

/* eigenvector hand-off for twin sectors (ed_twin): the state of sector B(ndw,nup) from the state of this
 * sector A(nup,ndw), vector_B(i) = cvec_A(Order(i)), Order = twin_sector_order (ED_SETUP.f90:852-880,
 * ED_EIGENSPACE.f90:486-497).  Vectors of getDim doubles, each either a host pointer or a device pointer (e.g. the
 * state edgpu_lanczos_gs_device left in HBM: nothing crosses PCIe then); out of place.  With more than one rank vin is
 * the rank's shard of A and vout its shard of B: DimDw_A * Qdw_B * DimPh doubles, Qdw_B = the rank's share of DimUp_A
 * under the dw split of ED_HAMILTONIAN.f90:96-110 (the state is all-gathered once on the device). */
int edgpu_twin_reorder(edgpu_handle h, const double *vin, double *vout);

/* ---- Green's-function leg: O|state> between two live sectors, in HBM ------
 * vout(j) = sgn * vin(i) with |j> = c^+_{iorb,ispin}|i> (dagger=1) or c_{iorb,ispin}|i> (dagger=0),
 * ED_GF_NORMAL.f90:190-213 / 264-283; `to` must be getCDGsector/getCsector of `from`
 * (ED_SETUP.f90:377-418).  vout is returned normalised, norm2 = <vout|vout> before normalisation
 * (the weight passed to add_to_lanczos_gf_normal, ED_GF_NORMAL.f90:239). */
int edgpu_apply_op(edgpu_handle from, edgpu_handle to, int32_t iorb, int32_t ispin, int32_t dagger,
                   const double *vin_dev, double *vout_dev, double *norm2);
/* the whole leg: apply O, normalise, sp_lanc_tridiag in the target sector.  `state_host` (here and in the chi /
 * phonon legs and edgpu_imp_density_offdiag) may be a host pointer or a device pointer (e.g. the eigenvector
 * edgpu_lanczos_gs_device left in HBM): device memory is used in place, host memory is uploaded once. */
int edgpu_gf_lanczos(edgpu_handle from, edgpu_handle to, int32_t iorb, int32_t ispin, int32_t dagger,
                     const double *state_host, int32_t nlanc, double threshold, double *norm2,
                     double *alanc, double *blanc, int32_t *nused);

/* susceptibility leg: vvinit_i = (sum_a cup[a] n_up,a + cdw[a] n_dw,a) state_i, normalised,
 * then sp_lanc_tridiag in the same sector.  Sz_a: cup=+1/2,cdw=-1/2 (ED_GF_CHISPIN.f90:169-171); n_a: 1,1
 * (ED_GF_CHIDENS.f90:166-168); mixed/total variants are sums of these (CHISPIN 282-284, 393-394).
 * cup/cdw: Norb doubles; norm2 is the weight passed to add_to_lanczos_spinChi/densChi. */
int edgpu_chi_lanczos(edgpu_handle h, const double *cup, const double *cdw, const double *state_host, int32_t nlanc,
                      double threshold, double *norm2, double *alanc, double *blanc, int32_t *nused);

/* phonon propagator leg: vvinit = (b + b^+)|state> (ED_GF_PHONON.f90:76-93), normalised, then sp_lanc_tridiag */
int edgpu_phonon_lanczos(edgpu_handle h, const double *state_host, int32_t nlanc, double threshold, double *norm2,
                         double *alanc, double *blanc, int32_t *nused);

/* Step-wise form of the same loop for an external clock (bench.py): `begin` sets up a fixed
 * start vector, `steps` enqueues iterations asynchronously on the library's stream (the caller
 * synchronises the device), `result` returns the mean product-kernel time measured with HIP
 * events on that stream and the first `ncoef` Lanczos coefficients (alpha_k, beta_{k+1}). */
int edgpu_lanczos_bench_begin(edgpu_handle h, int32_t max_steps);
int edgpu_lanczos_bench_steps(edgpu_handle h, int32_t nsteps, int32_t timed);
int edgpu_lanczos_bench_result(edgpu_handle h, int32_t nsteps_timed, float *hxv_ms_mean, double *alphas, double *betas, int32_t ncoef);
/* mean duration of the exchange step (pack + send/recv, or the callback) of the last timed batch; 0 on one rank */
int edgpu_lanczos_bench_comm_ms(edgpu_handle h, int32_t nsteps_timed, float *ms_mean);

/* ---- multi-GPU exchange (replaces MPI_AllGatherv/MPI_AllReduce of the MPI matvecs,
 * ED_SETUP.f90:689-733, ED_HAMILTONIAN_SPARSE_HxV.f90:674-692) --------------------------
 * RCCL: every rank calls unique_id on rank 0, the host broadcasts the 128 bytes, then init. */
int edgpu_comm_unique_id(void *id128);
int edgpu_comm_init_rccl(edgpu_handle h, const void *id128);
/* Host-supplied exchange (e.g. MPI or gloo), used when RCCL is not wanted.
 * allgatherv: send_dev[nsend] -> recv_dev at displs[r] (counts[r] doubles) for every rank r.
 * allreduce : in-place sum of n host doubles. */
typedef int (*edgpu_allgatherv_fn)(void *ctx, const double *send_dev, int64_t nsend, double *recv_dev,
                                   const int64_t *counts, const int64_t *displs, int32_t nranks);
typedef int (*edgpu_allreduce_fn)(void *ctx, double *buf_host, int32_t n);
/* exchange (all-to-all-v): sendcounts[r] doubles at send_dev+sdispls[r] go to rank r; recvcounts[r]
 * doubles from rank r land at recv_dev+rdispls[r].  Used for the "needed rows only" exchange: each
 * rank receives just the dw rows its own rows couple to (53/36/24 % of an all-gather at 2/4/8 ranks,
 * Ns=16).  allgatherv remains in use for sectors with Jx/Jp or phonon terms. */
typedef int (*edgpu_exchange_fn)(void *ctx, const double *send_dev, const int64_t *sendcounts, const int64_t *sdispls,
                                 double *recv_dev, const int64_t *recvcounts, const int64_t *rdispls, int32_t nranks);
int edgpu_comm_init_callbacks(edgpu_handle h, edgpu_allgatherv_fn ag, edgpu_exchange_fn xch, edgpu_allreduce_fn ar, void *ctx);

/* ---- the process-wide communicator ----------------------------------------------------------------------
 * The reference sets MpiComm ONCE (ed_set_MpiComm, ED_VARS_GLOBAL.f90:332-347) and every sector uses it; a DMFT
 * iteration builds O(Ns^2) sectors plus the GF target sectors.  Accordingly: create the communicator once per
 * process (RCCL: id from edgpu_comm_unique_id on rank 0, broadcast by the host; or host callbacks), attach it to
 * every sector after edgpu_sector_create, destroy it at the end.  The communicator must outlive the sectors it is
 * attached to.  edgpu_comm_init_rccl/_callbacks above remain as the one-communicator-per-sector convenience form. */
typedef struct edgpu_comm_s *edgpu_comm;
int edgpu_comm_create_rccl(const void *id128, int32_t rank, int32_t size, int32_t device /* -1: current */, edgpu_comm *out);
int edgpu_comm_create_callbacks(int32_t rank, int32_t size, edgpu_allgatherv_fn ag, edgpu_exchange_fn xch, edgpu_allreduce_fn ar,
                                void *ctx, edgpu_comm *out);
/* MEASUREMENT AID: a communicator that stands for rank `rank` of a `size`-rank job on ONE GPU.  A sector created with the same
 * mpi_rank / mpi_size and attached to it has the partition, row lists, x-space, packing kernels and stream structure of the real
 * job; what the peers would deliver is replaced by device-to-device copies of the same byte count on the comm stream, and the
 * all-reduces return the rank's own partial sums.  Timings and HBM traffic of one rank of P are real; the numbers computed are
 * not a result (bench.py --emulate-rank; ED_HAMILTONIAN.f90:96-110 is the partition it stands in for). */
int edgpu_comm_create_emulated(int32_t rank, int32_t size, edgpu_comm *out);
/* Sectors with fewer dw columns than ranks run on the first DimDw ranks (ED_HAMILTONIAN.f90:65-93: MPI_Group_incl +
 * MPI_Comm_create).  edgpu_comm_sub is that shrunk communicator: COLLECTIVE over `parent` (every rank calls it),
 * cached inside the parent (one split per distinct size); ranks outside get *out = NULL, *is_member = 0 and do not
 * create the sector.  RCCL communicators only; a host that owns its groups makes one callback communicator per group. */
int edgpu_comm_sub(edgpu_comm parent, int32_t nranks, edgpu_comm *out, int32_t *is_member);
int edgpu_comm_destroy(edgpu_comm c);
int edgpu_sector_attach_comm(edgpu_handle h, edgpu_comm c);
/* number of native (RCCL) communicators this process has created so far (tests: one per process, not per sector) */
int edgpu_comm_count(int32_t *n);
/* in-place sum of n host doubles over the ranks of the communicator */
int edgpu_comm_allreduce_sum(edgpu_comm c, double *buf_host, int32_t n);
/* what one product of this sector exchanges: mode 0 none, 1 needed dw rows (send/recv pairs), 2 all-gather, 3 distributed
 * transpose; send_bytes/recv_bytes (mpi_size entries each, may be NULL): bytes to / from every peer per product */
int edgpu_sector_exchange_info(edgpu_handle h, int32_t *mode, int64_t *send_bytes, int64_t *recv_bytes);
/* Routing of the needed-rows exchange (mode 1).  rows_direct[s*P+d]: dw rows rank s sends to rank d per product (the same
   matrix on every rank, derived from H_dw).  With EDGPU_RELAY=1 at sector creation (>= 3 ranks) the heavy pairs hand part of
   their rows to a third rank in a two-phase schedule: enabled, number of relayed chunks, rows on the busiest directed link
   before (one phase) and after (phase 1 + phase 2), rows_routed[a*P+b] = rows on link a->b under the plan.  Any output pointer
   may be NULL.  Replaces nothing in the reference (its exchange is MPI_AllGatherV, ED_SETUP.f90:689-733). */
int edgpu_sector_relay_info(edgpu_handle h, int32_t *enabled, int32_t *nchunks, int64_t *busiest_before, int64_t *busiest_after, int64_t *rows_direct, int64_t *rows_routed);

/* ---- sparse-matrix container (ED_SPARSE_MATRIX.f90): same semantics, product on the GPU ------------------
 * sp_init_matrix (127-172), sp_insert_element = duplicate-summing append in storage order (255-323),
 * sp_delete_matrix (184-237), sp_dump_matrix adds into a dense column-major Nrow x Ncol array (386-438).
 * Indices are 1-based as in the reference.  The sector operator does NOT go through this container (it
 * builds its factors directly); it is provided for host code that assembles other operators. */
typedef struct edgpu_spmat_s *edgpu_spmat;
int edgpu_sp_init_matrix(int64_t nrow, int64_t ncol /* 0 -> nrow */, edgpu_spmat *out);
/* MPI form: sp_set_mpi_matrix(istart,iend,ishift) followed by sp_init_matrix (ED_SPARSE_MATRIX.f90:574-585, 155-172):
 * the object holds the rows istart..iend (1-based, global) of an N x ncol matrix; insert/row take GLOBAL row
 * numbers and store at i - ishift (289-323); dump adds the local rows into a dense N x ncol array at their global
 * position (the reference then all-reduces, 408-438); matvec returns the iend-istart+1 local entries. */
int edgpu_sp_init_matrix_mpi(int64_t istart, int64_t iend, int64_t ishift, int64_t n, int64_t ncol /* 0 -> n */, edgpu_spmat *out);
/* sp_load_matrix (325-344): (re)initialise *inout from a dense column-major nrow x ncol array, inserting the
 * non-zeros row by row, columns ascending.  *inout may be NULL (new object) or an existing matrix (deleted first). */
int edgpu_sp_load_matrix(const double *dense, int64_t nrow, int64_t ncol, edgpu_spmat *inout);
int edgpu_sp_insert_element(edgpu_spmat m, double value, int64_t i, int64_t j);
int edgpu_sp_delete_matrix(edgpu_spmat m);
int edgpu_sp_dump_matrix(edgpu_spmat m, double *dense);
int edgpu_sp_row(edgpu_spmat m, int64_t i, int32_t *size, int64_t *cols, double *vals, int32_t cap);
/* Hv = A v (accumulate=0) or Hv += A v (accumulate=1); host vectors, product on the device */
int edgpu_sp_matvec(edgpu_spmat m, const double *v_host, double *hv_host, int32_t accumulate);
/* the same on device vectors (edgpu_malloc / any HIP allocation of ncol and nrow doubles): nothing crosses PCIe per call.
 * Asynchronous on `hip_stream` (a hipStream_t, NULL = the default stream); the caller synchronises. */
int edgpu_sp_matvec_device(edgpu_spmat m, const double *v_dev, double *hv_dev, int32_t accumulate, void *hip_stream);

/* The library keeps large work vectors (the stored Krylov basis of edgpu_lanczos_gs), small device blocks, the streams of
 * deleted sectors and the pinned host blocks of the Lanczos coefficients in process-wide caches between calls and sectors
 * (the host creates and deletes O(Ns^2) sectors per DMFT iteration); device memory is given back by itself when one of the
 * library's allocations fails.  Call this to return everything at once (e.g. before another library needs it). */
int edgpu_cache_release(void);

/* ---- device memory helpers for hosts without a HIP binding (Fortran) ---------------------- */
int edgpu_device_count(int32_t *n);
int edgpu_malloc(void **dev, size_t bytes);
int edgpu_free(void *dev);
int edgpu_memcpy_h2d(void *dev, const void *host, size_t bytes);
int edgpu_memcpy_d2h(void *host, const void *dev, size_t bytes);
int edgpu_memset(void *dev, int32_t byte_value, size_t bytes);   /* hipMemset: e.g. the all-zero start vector of ED_DIAG.f90:147 without a host copy */
int edgpu_device_synchronize(void);

const char *edgpu_last_error(void);
int edgpu_abi_version(void);

/* ---- environment switches ------------------------------------------------------------------
 * The integer-valued EDGPU_* environment variables this build of the library reads (at sector creation), with their
 * defaults (-1: decided by the library), a one-word kind ("form", "fallback", "multi-GPU", "test hook", "debug"), what they
 * select and the test that covers the non-default value.  Any other EDGPU_* name found in the environment is reported on
 * stderr once.  The reference has no counterpart (its switches are input-file variables, ED_INPUT_VARS.f90). */
int edgpu_knob_count(void);
int edgpu_knob_info(int32_t i, const char **name, int32_t *default_value, const char **kind, const char **doc, const char **covered_by);

#ifdef __cplusplus
}
#endif
#endif /* EDGPU_H */
