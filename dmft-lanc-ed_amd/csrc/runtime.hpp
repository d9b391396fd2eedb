// runtime.hpp -- error channel, RCCL loader, device buffers and the per-sector state behind edgpu_handle
// Part of libedgpu.so (one translation unit: edgpu.hip includes these headers in order).
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <algorithm>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/edgpu.h"
#include "sector_host.hpp"
#include "kernels.hpp"
#include "basis_kernels.hpp"

namespace edgpu {

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            int c_ = (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? EDGPU_ENODEVICE         \
                     : (e_ == hipErrorOutOfMemory ? EDGPU_ENOMEM : EDGPU_EHIP);                        \
            return fail(c_, std::string(#expr) + ": " + hipGetErrorString(e_));                       \
        }                                                                                              \
    } while (0)
#define CHK(expr) do { int r_ = (expr); if (r_ != EDGPU_OK) return r_; } while (0)

// Two copies of the HIP runtime in one process (e.g. this library bound to /opt/rocm's libamdhip64 and an embedding PyTorch
// that then maps the copy it ships) each initialise the driver and each run exit handlers over the other's state: such a
// process computes correctly and then dies at exit with "double free or corruption".  This library holds no static object
// whose destructor calls HIP (g_kpool, g_rccl are plain data), so the only cure is not to get into that state: every entry
// that would first touch the device refuses while more than one libamdhip64 is mapped.  The distinct paths are read from
// /proc/self/maps (≈0.2 ms; entries that create sectors / communicators / device memory only).
static int single_hip_runtime()
{
    // one entry per FILE (device:inode column of /proc/self/maps): the same library reached through two mounts, or unlinked
    // after it was mapped ("(deleted)"), is still one runtime
    std::vector<std::string> ids, paths;
    if (FILE *f = fopen("/proc/self/maps", "r")) {
        char line[4096];
        while (fgets(line, sizeof(line), f)) {
            if (!strstr(line, "libamdhip64")) continue;
            char dev[64] = "", path[4000] = ""; unsigned long long ino = 0;
            if (sscanf(line, "%*s %*s %*s %63s %llu %3999[^\n]", dev, &ino, path) < 2) continue;
            const std::string id = std::string(dev) + ":" + std::to_string(ino);
            bool dup = false;
            for (auto &s : ids) dup = dup || s == id;
            if (!dup) { ids.push_back(id); paths.push_back(path); }
        }
        fclose(f);
    }
    if (ids.size() <= 1) return EDGPU_OK;
    std::string msg = "edgpu: " + std::to_string(ids.size()) + " copies of the HIP runtime are mapped in this process (";
    for (size_t i = 0; i < paths.size(); i++) msg += (i ? ", " : "") + paths[i];
    msg += "): load the embedding framework (e.g. `import torch`) BEFORE libedgpu.so so that both bind to one runtime, or preload that "
           "runtime (LD_PRELOAD); a process that never loads such a framework sets EDGPU_HIP_RUNTIME=system in the Python binding; "
           "continuing would end in heap corruption at process exit";
    return fail(EDGPU_EHIP, msg);
}

static int require_device()
{
    CHK(single_hip_runtime());
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1) return fail(EDGPU_ENODEVICE, "edgpu: no HIP device visible (this library has no CPU fallback)");
    return EDGPU_OK;
}

// ------------------------------------------------------------------------------------------
// RCCL, resolved at run time so that the library loads on hosts without it and shares the
// copy an embedding process (e.g. PyTorch) may already have loaded.
// ------------------------------------------------------------------------------------------
struct NcclUid { char b[128]; };
struct Rccl {
    int (*GetUniqueId)(NcclUid *) = nullptr;
    int (*CommInitRank)(void **, int, NcclUid, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommSplit)(void *, int, int, void **, void *) = nullptr;     // optional (sub-communicators)
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
    bool load()
    {
        if (ok) return true;
        // Use the RCCL the process already has (an embedding PyTorch ships its own copy, built against the
        // HIP runtime it loaded); only then fall back to the system library.
        void *h = RTLD_DEFAULT;
        if (!dlsym(h, "ncclCommInitRank")) {
            h = nullptr;
            if (const char *e = getenv("EDGPU_RCCL_LIB")) h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
            if (!h) {
                if (FILE *f = fopen("/proc/self/maps", "r")) {
                    char line[4096];
                    while (!h && fgets(line, sizeof(line), f)) {
                        char *p = strstr(line, "librccl");
                        if (!p) continue;
                        char *path = strchr(line, '/');
                        if (!path) continue;
                        path[strcspn(path, "\n")] = 0;
                        h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
                    }
                    fclose(f);
                }
            }
            if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) return false;
        }
#define L(name) *(void **)(&name) = dlsym(h, "nccl" #name); if (!name) return false
        L(GetUniqueId); L(CommInitRank); L(CommDestroy); L(Send); L(Recv); L(AllReduce); L(GroupStart); L(GroupEnd); L(GetErrorString);
#undef L
        *(void **)(&CommSplit) = dlsym(h, "ncclCommSplit");
        ok = true;
        return true;
    }
};
static Rccl g_rccl;
constexpr int kNcclDouble = 8, kNcclSum = 0;

struct Comm {
    enum Kind { NONE, RCCL, CALLBACK, EMULATED } kind = NONE;      // EMULATED: one rank of `size` measured alone (edgpu_comm_create_emulated)
    int rank = 0, size = 1;
    void *nccl = nullptr;
    bool owned = false;      // the sector created `nccl` itself (edgpu_comm_init_rccl) and destroys it; attached communicators are not owned
    edgpu_allgatherv_fn ag = nullptr; edgpu_exchange_fn xch = nullptr; edgpu_allreduce_fn ar = nullptr; void *ctx = nullptr;
};
static int g_comm_created = 0;      // native (RCCL) communicators this process has created so far

// Process-wide cache of large work vectors (the stored Krylov basis of edgpu_lanczos_gs): allocating inside a running
// iteration costs ~30 ms per call on this runtime, so the buffers outlive the call and the sector and are handed to
// the next ground-state search; any allocation that fails gives them back first (kpool_release) and retries.
static void kpool_release();

// Small device blocks (tables, partial sums, work vectors of small sectors) are recycled through a process-wide pool of
// power-of-two size classes instead of going back to the driver: a sector of the everyday size (Ns <= 12) holds ~50 of them and
// the host builds and deletes O(Ns^2) sectors plus the Green's-function target sectors per DMFT iteration (ED_DIAG.f90:83-276);
// hipFree alone was ~4 ms per deleted sector (profiles/r04/sector_sweep_ns12_before.txt).  Blocks above 4 MiB go to the driver as
// before.  A block is handed out again only after the sector that held it has synchronised its device (destroy()) or, for
// temporaries, in stream order.  edgpu_cache_release() and any failed allocation return the pool to the driver.
struct SmallPool {
    static constexpr size_t kMaxBlock = (size_t)4 << 20, kMaxHeld = (size_t)1 << 30;
    std::map<std::pair<int, size_t>, std::vector<void *>> free_;      // (device, class bytes) -> blocks
    size_t held = 0;
    static size_t cls(size_t bytes) { size_t c = 256; while (c < bytes) c <<= 1; return c; }
    void *take(int dev, size_t c) { auto it = free_.find({dev, c}); if (it == free_.end() || it->second.empty()) return nullptr; void *q = it->second.back(); it->second.pop_back(); held -= c; return q; }
    bool give(int dev, size_t c, void *q) { if (held + c > kMaxHeld) return false; free_[{dev, c}].push_back(q); held += c; return true; }
    void drain() { for (auto &kv : free_) for (void *q : kv.second) (void)hipFree(q); free_.clear(); held = 0; }
};
static SmallPool g_spool;

// Streams are recycled the same way: on this runtime creating one costs ~0.6 ms and destroying it ~0.5 ms, and now and then a creation
// takes 70-160 ms (a new hardware queue) -- 0.37 s of a 1.2 s sweep over the 91 sectors of an Ns=12 model went into the two streams
// per sector (profiles/r04/sector_sweep_hip_api.txt).  A stream goes back only after it has been synchronised.
struct StreamPool {
    static constexpr size_t kMaxHeld = 16;
    std::map<int, std::vector<hipStream_t>> free_;
    int take(int dev, hipStream_t *out)
    {
        auto &v = free_[dev];
        if (!v.empty()) { *out = v.back(); v.pop_back(); return EDGPU_OK; }
        HIPCHK(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
        return EDGPU_OK;
    }
    void give(int dev, hipStream_t st)
    {
        if (!st) return;
        auto &v = free_[dev];
        if (v.size() < kMaxHeld && hipStreamSynchronize(st) == hipSuccess) v.push_back(st); else { (void)hipGetLastError(); (void)hipStreamDestroy(st); }
    }
    void drain() { for (auto &kv : free_) for (hipStream_t st : kv.second) (void)hipStreamDestroy(st); free_.clear(); }
};
static StreamPool g_streams;

template <class T> struct DevBuf {
    T *p = nullptr; size_t n = 0;
    size_t pooled = 0; int pdev = -1;      // size class and device of a pooled block (0: a driver allocation of its own)
    int alloc(size_t count)
    {
        release(); n = count; if (!count) return EDGPU_OK;
        const size_t bytes = count * sizeof(T);
#ifndef EDGPU_NO_POOL
        if (bytes <= SmallPool::kMaxBlock) {
            int dev = 0; (void)hipGetDevice(&dev);
            const size_t c = SmallPool::cls(bytes);
            void *q = g_spool.take(dev, c);
            if (!q) {
                hipError_t e = hipMalloc(&q, c);
                if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); kpool_release(); e = hipMalloc(&q, c); }
                if (e != hipSuccess) { p = nullptr; n = 0; }
                HIPCHK(e);
            }
            p = static_cast<T *>(q); pooled = c; pdev = dev;
            return EDGPU_OK;
        }
#endif
        hipError_t e = hipMalloc((void **)&p, bytes);
        if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); kpool_release(); e = hipMalloc((void **)&p, bytes); }
        if (e != hipSuccess) { p = nullptr; n = 0; }
        HIPCHK(e);
        return EDGPU_OK;
    }
    int upload(const std::vector<T> &h) { CHK(alloc(h.size())); if (h.size()) HIPCHK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return EDGPU_OK; }
    void release()
    {
        if (p) { if (!pooled || !g_spool.give(pdev, pooled, p)) (void)hipFree(p); }
        p = nullptr; n = 0; pooled = 0; pdev = -1;
    }
    ~DevBuf() { release(); }
};

// Lanczos coefficients (alphas, betas) live in pinned host memory that the device writes directly (fine-grained, mapped): a solver that
// looks at alpha_k / beta_k on the host after every step (edgpu_eigsh: the orthogonality estimates are kept on the host) then needs a
// stream synchronisation only.  Two 8-byte hipMemcpyAsync calls cost ~20 us each on the host side of this runtime -- as much as the four
// kernels of a small sector's iteration (profiles/r04/sector_sweep_hip_api.txt).  The kernels only WRITE these arrays (one lane,
// once per step; k_lanczos_wg reads one beta per launch); blocks are recycled through a process-wide pool like the device blocks.
struct PinnedPool {
    std::map<size_t, std::vector<void *>> free_;
    static size_t cls(size_t bytes) { size_t c = 4096; while (c < bytes) c <<= 1; return c; }
    void *take(size_t c) { auto it = free_.find(c); if (it == free_.end() || it->second.empty()) return nullptr; void *q = it->second.back(); it->second.pop_back(); return q; }
    void give(size_t c, void *q) { if (free_[c].size() < 64) free_[c].push_back(q); else (void)hipHostFree(q); }
    void drain() { for (auto &kv : free_) for (void *q : kv.second) (void)hipHostFree(q); free_.clear(); }
};
static PinnedPool g_pinned;
template <class T> struct CoefBuf {      // p: the device's view (kernel arguments), h: the host's view (read after a stream synchronisation)
    T *p = nullptr, *h = nullptr; size_t n = 0, cls = 0;
    int alloc(size_t count)
    {
        release(); if (!count) return EDGPU_OK;
        const size_t c = PinnedPool::cls(count * sizeof(T));
        void *q = g_pinned.take(c);
        if (!q) HIPCHK(hipHostMalloc(&q, c, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent));
        void *d = nullptr;
        if (hipHostGetDevicePointer(&d, q, 0) != hipSuccess || !d) { (void)hipGetLastError(); (void)hipHostFree(q); return fail(EDGPU_EHIP, "edgpu: pinned host memory is not mapped into the device address space"); }
        h = static_cast<T *>(q); p = static_cast<T *>(d); n = count; cls = c;
        return EDGPU_OK;
    }
    void release() { if (h) g_pinned.give(cls, h); p = h = nullptr; n = 0; cls = 0; }
    ~CoefBuf() { release(); }
};

// Single host thread per process (one rank per GPU, as the reference): the pool is not synchronised.  `in_use` is set for
// the duration of a ground-state search: a failing allocation elsewhere must not free the vectors that search is using.
struct KrylovPool { std::vector<double *> buf; std::vector<size_t> cap; int device = -1; double *slab = nullptr; size_t slab_cap = 0; bool in_use = false; };   // slab: all vectors of a small sector in one allocation
static KrylovPool g_kpool;
static void kpool_release()
{
    if (g_kpool.in_use) return;      // the out-of-memory retry then simply fails with EDGPU_ENOMEM
    for (double *q : g_kpool.buf) if (q) (void)hipFree(q);
    g_kpool.buf.clear(); g_kpool.cap.clear();
    if (g_kpool.slab) (void)hipFree(g_kpool.slab);
    g_kpool.slab = nullptr; g_kpool.slab_cap = 0;
    g_spool.drain();
}

struct DevFactor {
    DevBuf<int32_t> map, rank, rowptr, cols, ellc, talpha, tbeta;
    DevBuf<uint32_t> pcols, pcodes, pelem, pelem8, pelem16; DevBuf<double> ptval; bool packed = false, packed_elem = false, packed_elem8 = false, packed_elem16 = false; int32_t pterms = 0;      // pelem16: element-major image of 12 dwords (16 hop slots) for the one-configuration-per-lane kernels
    // segment + halo form of the local pass for rows beyond LDS (k_uppass_halo): the row is cut into nseg segments; seg_ext
    // lists, per segment and ascending, the columns outside it that its hops reach; pseg8 is pelem8 with every column
    // replaced by its position in the workgroup's LDS image [segment | halo]
    DevBuf<int32_t> seg_ext, seg_ext_off; DevBuf<uint32_t> pseg8; int32_t nseg = 0, seglen = 0, seg_maxhalo = 0; bool seg_ok = false;
    DevBuf<double> vals, ellv, E, tval, lin;
    int32_t width = 0;
    bool ragged = false;     // some row has fewer hops than `width` (ELL padding present)
};

}  // namespace edgpu

using namespace edgpu;

struct edgpu_sector_s {
    SectorHost H;
    Comm comm;
    int device = 0;
    hipStream_t stream = nullptr, cstream = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_gather = nullptr, ev_ready = nullptr;
    hipEvent_t lz_ev[2] = {nullptr, nullptr};      // ground-state search on launch-bound sectors: end of the chunk of iterations being judged / of the one issued ahead
    std::vector<std::unique_ptr<DevFactor>> fac;
    // the two factors the product kernels run on: fac[0], fac[1] (ed_total_ud=T) or the composite blocks (ed_total_ud=F, cdev)
    std::unique_ptr<DevFactor> cdev[2]; DevFactor *upf = nullptr, *dwf = nullptr; const Factor *hup = nullptr, *hdw = nullptr;
    bool kron2 = false, composite = false;
    // Lanczos vectors of the fused iteration live with a row pitch padded to the 128-B line (vld doubles, vlen = rows * vld);
    // vld == DimUp when padding is off.  Vectors the caller passes in and out keep the reference's dense layout.
    int64_t vld = 0, vlen = 0; bool pad_on = false;
    DevBuf<double> diag, cross, gup, gdw, bd_up, bd_dw;
    DevBuf<int32_t> stride;
    // explicit CSR
    DevBuf<int64_t> csr_rowptr, uppre, dwpre; DevBuf<int32_t> csr_cols; DevBuf<double> csr_vals;
    DevBuf<int64_t> csrD_rowptr; DevBuf<int32_t> csrD_cols; DevBuf<double> csrD_vals;   // dw block of the split form
    // every row of one idw row holds the same number of dw hops: the dw-type blocks keep their row pointer in the factored
    // form rowptr(iup, idw) = base[idw] + iup * cnt[idw] (the full arrays above stay for the row export only) and the
    // product streams no row pointer for them
    DevBuf<int64_t> csrD_base, csrO_base; DevBuf<int32_t> csrD_cnt, csrO_cnt; bool csr_uni = false;
    // lane-major image of the 64-row blocks (k_csr_il): the local block when every up configuration has the same number of
    // hops, the dw-type blocks always (their rows are uniform per idw row)
    bool csr_il_loc = false, csr_il_dw = false; bool csrL_nt = false;
    bool packed_ahead = false;      // hxv has packed the send rows on the compute stream already (halo_pack): halo_exchange does not pack again
    bool x_own_in_v = false;  // several ranks, stored-factor / on-the-fly forms with the needed-rows exchange: the dw pass reads the rank's own rows from the input vector, the x-space only holds received rows
    // self-reducing Lanczos iteration of launch-bound sectors (k_kron2<KF_SR> + k_axpy_norm_sr): two launches per iteration
    bool sr_on = true;        // EDGPU_LZ_SR
    bool sr_allow = false;    // set by the plain Lanczos loops around their iterations (the eigensolvers interleave scalar steps of their own)
    bool sr_now = false;      // this product is the first half of such an iteration
    bool sr_pending = false;  // beta^2 partial sums of the last axpy pass not reduced yet
    int sr_nb = 0, sr_na = 0; DevBuf<double> partial_b;
    bool dvd_fused_on = true; // EDGPU_DVD_FUSED: Davidson step in two passes over the space (k_dvd_step + k_dvd_orth)
    bool lz_wg_on = true;     // EDGPU_LZ_WG: sectors that fit LDS run their Lanczos loops in one workgroup (k_lanczos_wg)
    DevBuf<int32_t> wg_done; DevBuf<double> wg_slab; DevBuf<uint32_t> wg_brec;      // iteration count written by that kernel; its Krylov slab
    // single rank: the stored columns index vectors whose rows are padded to the 128-B line (csr_ld doubles per row; == DimUp
    // when off); the Lanczos loops then run on pitched vectors, a caller's dense vectors go through csr_xs / csr_ys
    int axpy_nb = 512;      // workgroups of the axpy+norm pass of the three-pass iteration
    int64_t csr_ld = 0; DevBuf<double> csr_xs, csr_ys;
    DevBuf<double> crow_tab;      // DIRECT, rows beyond LDS: k_crow_table
    bool csr_tm = false; DevBuf<int64_t> csrD_cpre, csrO_cpre; std::vector<int32_t> h_cntD, h_cntO;      // tile-major order of the dw-type blocks
    // needed-rows exchange: the dw block is stored as two blocks, hops into the rank's own rows (runs while the exchange
    // is in flight) and hops into received rows (csrD_*: runs after it)
    DevBuf<int64_t> csrO_rowptr; DevBuf<int32_t> csrO_cols; DevBuf<double> csrO_vals; bool csr_three = false; int64_t csrO_nnz = 0;
    // non-local (Jx/Jp) block of the explicit storage, compressed rows: k_nd_build / k_csr_nd
    DevBuf<double> csr_gel;      // phonon sectors: electronic factor of the e-ph coupling (the reference's diagonal spH0e_eph)
    DevBuf<int64_t> nd_rows, nd_ptr; DevBuf<int32_t> nd_cols; DevBuf<double> nd_vals; DevBuf<uint8_t> nd_slot; int64_t nd_nr = 0, nd_nnz = 0;
    bool csr_split = false; int64_t csrL_nnz = 0, csrD_nnz = 0; int32_t csrD_rpb = 64, csrD_bpr = 0; int csrD_nt = 1; bool csrL_small = false, csrD_small = false;
    int64_t csr_nnz = 0; int32_t csr_rpb = 0, csr_bpr = 0, csr_tile = 1;
    int csr_xcd_rows = 0;
    // work vectors
    // halo exchange (multi-rank, ed_total_ud=T without Jx/Jp/phonon extras): x-space of the dw hops is
    // [own rows | rows needed from rank 0 | rank 1 | ...] instead of the whole gathered vector
    bool halo = false; int64_t nhalo = 0;
    std::vector<int64_t> send_rows_cnt, recv_rows_cnt, send_rows_off, recv_rows_off;
    DevBuf<int32_t> send_list, dw_ellc_x, dw_cols_x, xmap_dev; DevBuf<double> sendbuf;
    hipEvent_t ev_pack = nullptr;
    // Relay routing of that exchange (opt-in, EDGPU_RELAY=1, >= 3 ranks): the rows of a heavy pair travel partly through a third
    // rank in two phases (relay_plan, exchange.hpp).  xch_rows[s*P+d]: rows rank s sends to rank d per product, on every rank.
    struct RelayPiece { int64_t off = 0, cnt = 0; };       // rows: offset in the buffer it is read from / written to, and count
    struct RelayPeer { RelayPiece s1d, s1r, r1d, r1r, s2d, s2f, r2d, r2f; };   // s/r: send/receive, 1/2: phase, d: direct rows, r: rows handed to a relay, f: forwarded rows
    std::vector<int64_t> xch_rows, relay_link_rows;        // relay_link_rows[a*P+b]: rows on link a->b, both phases, under the plan
    std::vector<RelayPeer> relay; bool relay_on = false; int relay_chunks = 0; int64_t relay_before = 0, relay_after = 0; DevBuf<double> relay_stage;
    std::vector<std::unique_ptr<DevBuf<double>>> eigV; DevBuf<double> eig_h, eig_y, eig_partial;
    CoefBuf<double> eig_hp;      // one rank: dots and coefficients of the re-orthogonalisation passes travel through pinned host memory the kernels read and write (no copy calls)
    DevBuf<double> vfull, tmp_in, tmp_out, lz[5], scal, zvec, partial, partial2;
    CoefBuf<double> alphas, betas;      // pinned host memory written by the device (see CoefBuf)
    CoefBuf<double> partial_bh; bool sr_host_partials = false;      // edgpu_eigsh: host copy of the pending beta^2 partial sums of the self-reducing iteration
    int64_t nloc = 0, nloc_el = 0, vecdim = 0;
    int32_t nmax = 0;
    bool use_lds = false; int kron_threads = 256, kron_threads_override = 0; size_t kron_lds = 0;
    bool two_factor = false, extra = false, has_ph = false, has_nd = false, multi = false, kron_split = false; int dw_w = 64, dw_variant = 2, dw_rows = 16; bool up_persistent = true, up_pairs = true; int up_grid = 256, up_threads = 1024, row_split = 4; bool up_big = true, up_halo = true, nd_split = true;
    std::vector<hipEvent_t> bench_ev; int bench_k = 0; bool bench_fused = false; double *bv[3] = {nullptr, nullptr, nullptr};
    // distributed-transpose form of the dw hops (opt-in, EDGPU_DW_TRANSPOSE=1)
    bool tr_on = false; int64_t Qup = 0, up0 = 0; DevBuf<double> tA, tB, tVT, tHVT; bool tr_lds = false;
    // chunked pipeline of that exchange: the all-to-all of up-block chunk c+1 travels while chunk c runs its row pass (own stream)
    int tr_chunks = 4; hipStream_t tstream = nullptr; std::vector<hipEvent_t> tr_ev;
    std::vector<hipEvent_t> comm_ev; int comm_slot = -1;     // timed bench steps: events around the exchange on the comm stream
    int64_t nnz_total = 0, bytes_stored = 0;
    int64_t npartial = 0;
    // knobs read from the environment ONCE, at sector creation (never inside a product)
    bool direct_built = false, direct_cache = true, seg_remapped = false;      // on-the-fly tables derived once per sector (direct_tables)
    bool up_elem16 = true;    // one-configuration-per-lane local passes read the element-major 12-dword image (EDGPU_UP_ELEM16=0: the slot-major one)
    bool pad_allowed = true;  // EDGPU_PAD != 0: work vectors of the Lanczos loops may carry a padded row pitch
    bool up_pairs10 = true;   // pair form of the LDS-row local passes for 9-10 up hops (EDGPU_UP_PAIRS10=0: one configuration per lane, as before round 3)
    int dbg = 0; bool tables_wave = true; int dw_lds_kb = 0; bool lz_fused_on = true;
    int eig_ncv_used = 0;     // basis size the last edgpu_eigsh call actually ran with
    // Lanczos iteration fusion (single rank)
    // several ranks: one 16-byte all-reduce per Lanczos iteration instead of two of 8 bytes (EDGPU_ONE_ALLREDUCE=1; lz_iter_deferred)
    bool one_ar = false, one_ar_on = false, npend = false, defer_ar = false; int npend_k = 0;
    int fuse_mode = -1; const double *fuse_zvec = nullptr;   // >= 0: the product's final reduction also performs this scalar step
    edgpu_sector_s() = default;
    edgpu_sector_s(const edgpu_sector_s &) = delete;
    edgpu_sector_s &operator=(const edgpu_sector_s &) = delete;
    ~edgpu_sector_s();      // also runs when edgpu_sector_create fails half-way: nothing leaks
};

// The process-wide communicator of the host (the reference sets MpiComm once, ED_VARS_GLOBAL.f90:332-347): created once,
// attached to every sector (edgpu_sector_attach_comm).  `subs` caches the shrunk communicators of sectors with fewer dw
// columns than ranks (ED_HAMILTONIAN.f90:65-93), keyed by their size.
struct edgpu_comm_s {
    edgpu::Comm c;
    int device = 0;
    std::map<int, edgpu_comm_s *> subs;
    bool member = true;      // false: this rank is outside the (sub-)communicator
};

inline edgpu_sector_s::~edgpu_sector_s()
{
    if (comm.kind == Comm::RCCL && comm.nccl && comm.owned && g_rccl.ok) g_rccl.CommDestroy(comm.nccl);
    for (auto e : bench_ev) (void)hipEventDestroy(e);
    for (auto e : comm_ev) (void)hipEventDestroy(e);
    if (ev_pack) (void)hipEventDestroy(ev_pack);
    if (ev_a) (void)hipEventDestroy(ev_a);
    if (ev_b) (void)hipEventDestroy(ev_b);
    if (ev_gather) (void)hipEventDestroy(ev_gather);
    if (ev_ready) (void)hipEventDestroy(ev_ready);
    for (auto e : lz_ev) if (e) (void)hipEventDestroy(e);
    for (auto e : tr_ev) (void)hipEventDestroy(e);
    g_streams.give(device, tstream); g_streams.give(device, stream); g_streams.give(device, cstream);
}

namespace edgpu {
using S = edgpu_sector_s;

static int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
// launch attributes (dynamic LDS limit) are set once per function AND device
struct OncePerDevice { bool done[64] = {}; bool first(int dev) { if (dev < 0 || dev >= 64) return true; const bool f = !done[dev]; done[dev] = true; return f; } };

// ---------------------------------------------------------------------------------------
// Environment switches.  The shipping library reads exactly the names of this table (edgpu_knob_count / edgpu_knob_info export
// it; tests/test_knobs_cpu.py checks it against the call sites, against scripts/README.md and against the strings of the
// binary): forms whose choice awaits a multi-GPU measurement, fallbacks that a test exercises, and the hooks the tests use to
// drive large-sector code paths on small sectors.  Every other tuning value is a compile-time constant (XKNOB) that becomes an
// environment switch only in the experiments build (build.sh exp: -DEDGPU_EXPERIMENTS, libedgpu_exp.so, used by scripts/).
// A value of -1 means "decided by the library" (size thresholds).  Unknown EDGPU_* names in the environment are reported once.
// ---------------------------------------------------------------------------------------
struct KnobDef { const char *name; int dflt; const char *kind; const char *doc; const char *test; };
static const KnobDef kKnobs[] = {
    // product forms
    {"EDGPU_KRON_SPLIT", -1, "form", "two-pass product (local pass + column-tiled dw pass) of the stored-factor / on-the-fly operator; -1: once a vector exceeds 12 MiB", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_CSR_SPLIT", -1, "form", "explicit CSR stored as two blocks H_loc + H_dw; -1: once a vector exceeds 12 MiB", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_COMPOSITE", 1, "fallback", "ed_total_ud=F through composite up/dw blocks; 0: generic mixed-radix kernel", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_ND_SPLIT", 1, "fallback", "Jx/Jp terms in a pass of their own; 0: inside the row-wise kernel", "test_gpu_parity.py::test_c5_full_size_properties"},
    {"EDGPU_LZ_FUSED", 1, "fallback", "fused two-pass Lanczos iteration; 0: product + axpy/norm pass", "test_gpu_parity.py::test_fused_iteration_equals_three_pass_iteration"},
    {"EDGPU_PAD", 1, "fallback", "Lanczos work vectors with the row pitch padded to the 128-B line; 0: dense rows", "test_gpu_parity.py::test_fused_iteration_equals_three_pass_iteration"},
    {"EDGPU_ODD_PAIRS", 1, "fallback", "column-pair kernels on pitched vectors for odd DimUp; 0: scalar forms", "test_gpu_parity.py::test_odd_row_length_uses_the_pair_kernels_on_pitched_vectors"},
    {"EDGPU_UP_PAIRS10", 1, "fallback", "pair form of the local pass for 9-10 up hops; 0: one configuration per lane", "test_gpu_parity.py::test_pair_form_with_ten_hop_slots"},
    {"EDGPU_UP_PERSISTENT", 1, "fallback", "persistent LDS-row local pass; 0: row-wise kernel without the dw hops", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_UP_PAIRS", 1, "fallback", "two up configurations per lane in the local pass; 0: one", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_UP_THREADS", 1024, "fallback", "threads of the one-configuration-per-lane local pass (1024 or 512)", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_UP_GRID", 256, "test hook", "workgroups of the persistent local passes (tests: idle and odd counts)", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_UP_HALO", 1, "fallback", "segment + halo local pass for rows beyond LDS; 0: gather kernel k_uppass_big", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_UP_BIG", 1, "fallback", "packed-image gather kernel for rows beyond LDS; 0: row-wise kernel", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_UP_SEGMAX", 4096, "test hook", "longest row segment of the segment + halo local pass (tests: many short segments)", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_ROW_SPLIT", 4, "fallback", "workgroups per row of the gather kernel for rows beyond LDS", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_DW_W", 64, "test hook", "column-tile width of the dw pass (tests: narrow tiles on small sectors)", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_DW_ROWS", 16, "test hook", "rows per workgroup of the dw pass", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_DW_VARIANT", 2, "fallback", "dw pass: 2 two columns per lane, 3 the same with non-temporal result accesses, 1 one column per lane", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    // explicit CSR
    {"EDGPU_CSR_IL", 3, "fallback", "lane-major 64-row blocks: bit 0 local block, bit 1 dw-type blocks; 0: row-major blocks staged through LDS", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_CSR_TM", 1, "fallback", "tile-major storage of the lane-major dw-type blocks", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_CSR_JAG", 1, "fallback", "lane-major local block also when its rows differ in length (level-packed)", "test_gpu_parity.py::test_split_csr_with_rows_of_different_lengths"},
    {"EDGPU_CSR_PITCH", 1, "fallback", "columns in the pitched vector space on one rank", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_CSR_UNI", 1, "fallback", "factored row pointer of the dw-type blocks", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_CSRD_ROWS", 64, "test hook", "rows per block of the row-major dw block", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    {"EDGPU_CSRD_NT", 1, "fallback", "non-temporal entry loads of the dw block", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
    // Lanczos loops and eigensolvers
    {"EDGPU_LZ_STORE", 1, "fallback", "keep the Krylov vectors of a ground-state search in HBM; 0: second Lanczos pass", "test_gpu_parity.py::test_ground_state_with_stored_krylov_vectors"},
    {"EDGPU_LZ_STORE_MAX", 1 << 30, "test hook", "cap on the stored Krylov vectors (tests: running out mid-search)", "test_gpu_parity.py::test_ground_state_with_stored_krylov_vectors"},
    {"EDGPU_LZ_SR", 1, "fallback", "launch-bound sectors (whole product in one kernel, up to 4 M states): two launches per Lanczos iteration, every workgroup sums the partial sums itself; 0: four launches", "test_small_sector_lanczos.py::test_self_reducing_iteration_equals_the_four_launch_loop"},
    {"EDGPU_LZ_WG", 1, "fallback", "sectors of up to ~8 000 states run their Lanczos loops (sp_lanc_eigh, sp_lanc_tridiag and the legs built on it) in one workgroup; 0: per-iteration launches", "test_small_sector_lanczos.py::test_workgroup_lanczos_equals_the_launch_per_iteration_loop"},
    {"EDGPU_EIGSH_PRO", 1, "fallback", "thick-restart Lanczos with partial re-orthogonalisation; 0: whole-basis form", "test_eigsh.py::test_eigsh_alternative_forms"},
    {"EDGPU_EIGSH_FUSED", 1, "fallback", "fused two-pass products inside edgpu_eigsh", "test_eigsh.py::test_eigsh_alternative_forms"},
    {"EDGPU_EIGSH_OMEGA_PCT", 100, "test hook", "re-orthogonalisation trigger in % of sqrt(eps) (tests: always / never)", "test_eigsh.py::test_eigsh_alternative_forms"},
    {"EDGPU_EIGSH_MEM_MB", 0, "test hook", "pretend only this much device memory is free for the basis", "test_eigsh.py::test_eigsh_runs_with_the_basis_that_fits"},
    {"EDGPU_DVD_FUSED", 1, "fallback", "Davidson, spaces of up to 12 vectors: two passes over the space per step (residual, correction, Gram-Schmidt dots, new column of the projected matrix and the restart rotation in one; normalised new basis vector in the other); 0: separate passes and rotation kernels", "test_eigsh.py::test_davidson_two_pass_step_against_separate_passes"},
    {"EDGPU_ROT2", 1, "fallback", "restart rotation with 16-byte accesses on contiguous pieces", "test_eigsh.py::test_eigsh_alternative_forms"},
    {"EDGPU_ROT_MFMA", 1, "fallback", "restart rotation on the f64 matrix cores; 0: register-resident kernels", "test_eigsh.py::test_eigsh_alternative_forms"},
    // several ranks: forms awaiting a measurement on xGMI
    {"EDGPU_HALO", 1, "multi-GPU", "needed-rows exchange; 0: all-gather of the input vector", "test_multirank.py::test_multirank_fallback_forms"},
    {"EDGPU_DW_TRANSPOSE", 0, "multi-GPU", "dw hops through the distributed transpose (the reference's MPI scheme)", "test_multirank.py::test_multirank_distributed_transpose"},
    {"EDGPU_TR_CHUNKS", 4, "multi-GPU", "pipeline chunks of the transpose exchange", "test_multirank.py::test_multirank_distributed_transpose"},
    {"EDGPU_ONE_ALLREDUCE", 0, "multi-GPU", "deferred norm: one 16-byte all-reduce per Lanczos iteration", "test_multirank.py::test_multirank_one_allreduce_per_iteration"},
    {"EDGPU_RELAY", 0, "multi-GPU", "two-phase relay routing of the needed-rows exchange (>= 3 ranks)", "test_multirank.py::test_multirank_relay_routing"},
    {"EDGPU_CSR_THREE", 1, "multi-GPU", "explicit CSR of several ranks in three blocks (own-row dw hops overlap the exchange)", "test_multirank.py::test_multirank_fallback_forms"},
    {"EDGPU_CSR_PITCH_MULTI", 1, "multi-GPU", "pitched x-space and messages for the explicit CSR of several ranks", "test_multirank.py::test_multirank_fallback_forms"},
    // diagnostics
    {"EDGPU_DEBUG", 0, "debug", "bit 4: no LDS staging of rows (forces the long-row kernels on small sectors; tests), bit 8: solver statistics on stderr", "test_gpu_parity.py::test_large_sector_code_paths_on_small_sectors"},
};
static const char *const kKnobStrings[] = {"EDGPU_RCCL_LIB", "EDGPU_HIP_RUNTIME", "EDGPU_LIBRARY"};      // path-valued names (the last two are read by the Python binding)
constexpr int kNumKnobs = (int)(sizeof(kKnobs) / sizeof(kKnobs[0]));
static const KnobDef *knob_find(const char *name)
{
    for (int i = 0; i < kNumKnobs; i++) if (strcmp(kKnobs[i].name, name) == 0) return &kKnobs[i];
    return nullptr;
}
// value of a table knob (its default when the variable is unset)
static int knob(const char *name)
{
    const KnobDef *k = knob_find(name);
    if (!k) { fprintf(stderr, "edgpu: internal: environment switch %s is not in the table\n", name); abort(); }
    const char *e = getenv(name);
    return e ? atoi(e) : k->dflt;
}
#ifdef EDGPU_EXPERIMENTS
static int xknob_env(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
#define XKNOB(name, dflt) xknob_env(name, dflt)
#else
#define XKNOB(name, dflt) (dflt)      // shipping build: a constant, the name does not reach the binary
#endif
static int dbg_flags() { return knob("EDGPU_DEBUG"); }
extern "C" char **environ;
static void warn_unknown_knobs()
{
    static bool done = false;
    if (done) return;
    done = true;
    for (char **e = environ; e && *e; e++) {
        if (strncmp(*e, "EDGPU_", 6) != 0) continue;
        const char *eq = strchr(*e, '=');
        const std::string name(*e, eq ? (size_t)(eq - *e) : strlen(*e));
        bool known = knob_find(name.c_str()) != nullptr;
        for (const char *sname : kKnobStrings) known = known || name == sname;
#ifdef EDGPU_EXPERIMENTS
        known = true;
#endif
        if (!known) fprintf(stderr, "edgpu: warning: environment variable %s is not a switch of this library (edgpu_knob_info lists them; tuning switches exist in the experiments build only)\n", name.c_str());
    }
}

}  // namespace edgpu
