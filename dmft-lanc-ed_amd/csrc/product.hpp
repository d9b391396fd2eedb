// product.hpp -- hxv(): launches of one sector product for every storage / partition / epilogue variant
// Part of libedgpu.so (one translation unit: edgpu.hip includes these headers in order).
#pragma once

namespace edgpu {

// ---------------------------------------------------------------------------------------
// one product  out = H*v  (epi=false)   or the Lanczos-fused form (epi=true):
//   out = s_cur*H*u - beta*s_prev*u_prev ;  scal[TMP] = (s_cur*u) . out
// ---------------------------------------------------------------------------------------
// scal[TMP] = sum(partial[0..n)) : one pass for short arrays, two passes for long ones
static int reduce_partials(S *s, int64_t n)
{
    const double *src = s->partial.p;
    if (n > 16384) {
        const int64_t chunk = cdiv(n, 1024);
        const int nb = (int)cdiv(n, chunk);
        hipLaunchKernelGGL(k_reduce_chunks, dim3(nb), dim3(256), 0, s->stream, src, n, chunk, s->partial2.p);
        src = s->partial2.p; n = nb;
    }
    if (s->fuse_mode >= 0) hipLaunchKernelGGL(k_reduce_final_step, dim3(1), dim3(1024), 0, s->stream, src, n, s->scal.p, s->alphas.p, s->betas.p, s->fuse_zvec, s->fuse_mode);
    else hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s->stream, src, n, s->scal.p + SC_TMP);
    HIPCHK(hipGetLastError());
    return EDGPU_OK;
}

template <int F> static void launch_kron2(S *s, const Kron2Args &A, int64_t rows)
{
    size_t lds = (F & KF_LDS) ? s->kron_lds : 24 * sizeof(double);
    static OncePerDevice once;
    if (lds > 64 * 1024 && once.first(s->device)) (void)hipFuncSetAttribute((const void *)k_kron2<F>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_kron2<F>, dim3((unsigned)rows), dim3(s->kron_threads), lds, s->stream, A);
}

static int dispatch_kron2(S *s, int flags, const Kron2Args &A, int64_t rows)
{
#define CASE(f) case (f): launch_kron2<(f)>(s, A, rows); break;
#define CASE4(b) CASE(b) CASE((b) | KF_DIRECT) CASE((b) | KF_EPI) CASE((b) | KF_DIRECT | KF_EPI)
#define CASE8(b) CASE4(b) CASE4((b) | KF_PH | KF_ND)
#define CASESR(b) CASE((b) | KF_EPI | KF_SR) CASE((b) | KF_DIRECT | KF_EPI | KF_SR)
    switch (flags) {
        CASE8(0) CASE8(KF_LDS) CASE8(KF_NODW) CASE8(KF_NODW | KF_LDS) CASE8(KF_NOLOCAL)
        CASESR(0) CASESR(KF_LDS) CASESR(KF_PH | KF_ND) CASESR(KF_LDS | KF_PH | KF_ND)      // self-reducing Lanczos iteration (whole product in one kernel only)
    default: return fail(EDGPU_EINVAL, "edgpu: internal kernel variant missing");
    }
#undef CASESR
#undef CASE8
#undef CASE4
#undef CASE
    HIPCHK(hipGetLastError());
    return EDGPU_OK;
}

// The tables of the on-the-fly form (hop images of the factors from the bit patterns, row constants of the diagonal) depend on
// the sector descriptor only, which never changes: derived by the first product and kept (EDGPU_DIRECT_CACHE=0: derived again by
// every product, as the reference's directMatVec re-derives every element; three short launches, 45 us -- 2 % of a product at Ns=16,
// a fifth of a Lanczos iteration at Ns=12).  What the form does not hold is the O(N) diagonal.
static int direct_tables(S *s)
{
    if (s->direct_built && s->direct_cache) return EDGPU_OK;
    const SectorHost &H = s->H; const edgpu_desc &d = H.desc;
    auto crow_table = [&]() -> int {
        if (s->kron2 && s->dwf && s->dwf->E.p && s->dwf->map.p && s->cross.p) {
            // row constants of the on-the-fly diagonal, Edw + econst + cross(up impurity word, dw impurity word), per dw configuration
            // (two-factor and composite sectors alike: the local-pass kernels read them a row ahead instead of chasing mapdw -> cross)
            const int32_t nI = 1 << d.Norb; const int64_t nct = H.DimDw * nI;
            if (s->crow_tab.n != (size_t)nct) CHK(s->crow_tab.alloc((size_t)nct));
            hipLaunchKernelGGL(k_crow_table, dim3((unsigned)cdiv(nct, 256)), dim3(256), 0, s->stream, (const double *)s->dwf->E.p, (const int32_t *)s->dwf->map.p, (const double *)s->cross.p, H.econst, (int32_t)(nI - 1), nI, H.DimDw, s->crow_tab.p);
            HIPCHK(hipGetLastError());
        }
        return EDGPU_OK;
    };
    if (s->composite) { CHK(crow_table()); s->direct_built = true; return EDGPU_OK; }      // composite blocks (ed_total_ud=F) keep their tables; only the diagonal is formed on the fly
    for (int k = 0; k < 2 * H.Ns_Ud; k++) {
        DevFactor &f = *s->fac[k];
        TabArgs T{};
        T.dim = H.fac[k].dim; T.nterms = (int32_t)H.fac[k].terms.size(); T.width = f.width; T.transposed = (k < H.Ns_Ud || s->tr_on) ? 1 : 0;
        T.map = f.map.p; T.rank = f.rank.p; T.talpha = f.talpha.p; T.tbeta = f.tbeta.p; T.tval = f.tval.p;
        T.ellc = f.ellc.p; T.ellv = f.ellv.p; T.lin = f.lin.p; T.nlev = H.fac[k].nbits;
        T.pcols = f.packed ? f.pcols.p : nullptr; T.pcodes = f.packed ? f.pcodes.p : nullptr; T.pelem = f.packed_elem ? f.pelem.p : nullptr; T.pelem8 = f.packed_elem8 ? f.pelem8.p : nullptr; T.pelem16 = f.packed_elem16 ? f.pelem16.p : nullptr;
        T.impmask = d.ed_total_ud ? (1 << d.Norb) - 1 : 0; T.pair = d.ed_total_ud ? d.Ust - d.Jh : 0.0; T.E = f.E.p;
        const auto &tm = H.fac[k].terms;
        bool wave_ok = tm.size() <= 64 && f.width <= 64;
        for (size_t a = 0; a < tm.size() && wave_ok; a++)
            for (size_t b = a + 1; b < tm.size(); b++) if (tm[a].alpha == tm[b].alpha && tm[a].beta == tm[b].beta) { wave_ok = false; break; }
        if (wave_ok && s->tables_wave) hipLaunchKernelGGL(k_direct_tables_wave, dim3((unsigned)cdiv(T.dim, 4)), dim3(256), 0, s->stream, T);
        else if (f.width <= 16) hipLaunchKernelGGL(k_direct_tables<16>, dim3((unsigned)cdiv(T.dim, 64)), dim3(64), 0, s->stream, T);
        else hipLaunchKernelGGL(k_direct_tables<64>, dim3((unsigned)cdiv(T.dim, 64)), dim3(64), 0, s->stream, T);
    }
    HIPCHK(hipGetLastError());
    CHK(crow_table());      // reads the per-spin energy table the loop above has just derived
    s->direct_built = true;
    return EDGPU_OK;
}

// dw hops through the distributed transpose (see k_transpose2d): v (local shard) -> ... -> s->tA = (H_dw (x) 1) v, transposed
static int transposed_dw_pass(S *s, const double *v, hipStream_t st)
{
    const SectorHost &H = s->H; Comm &c = s->comm;
    const int P = c.size;
    auto split = [&](int64_t D, int r, int64_t &q, int64_t &d0) { const int64_t qq = D / P, rem = D % P; q = r < rem ? qq + 1 : qq; d0 = r < rem ? r * (qq + 1) : r * qq + rem; };
    std::vector<int64_t> sc(P), sd(P), rc(P), rd(P);
    TBlkArgs Bk{}; Bk.P = P; Bk.Qup = s->Qup; Bk.DimDw = H.DimDw;
    if (P > 64) return fail(EDGPU_EUNSUPPORTED, "edgpu: transpose exchange supports up to 64 ranks");
    for (int r = 0; r < P; r++) {
        int64_t qu, u0, qd, d0; split(H.DimUp, r, qu, u0); split(H.DimDw, r, qd, d0);
        sc[r] = qu * H.Qdw; sd[r] = H.Qdw * u0;          // rows iup of rank r's up-block out of my transposed shard
        rc[r] = s->Qup * qd; rd[r] = s->Qup * d0;        // my up-block out of rank r's dw columns
        Bk.dw0[r] = d0; Bk.dw0[r + 1] = d0 + qd;
    }
    auto a2a = [&](const double *send, const std::vector<int64_t> &scn, const std::vector<int64_t> &sdp, double *recv, const std::vector<int64_t> &rcn, const std::vector<int64_t> &rdp) -> int {
        HIPCHK(hipMemcpyAsync(recv + rdp[c.rank], send + sdp[c.rank], (size_t)scn[c.rank] * 8, hipMemcpyDeviceToDevice, st));
        if (P == 1) return EDGPU_OK;
        if (c.kind == Comm::RCCL) {
            int rc2 = g_rccl.GroupStart();
            for (int r = 0; r < P && rc2 == 0; r++) {
                if (r == c.rank) continue;
                if (scn[r]) rc2 = g_rccl.Send(send + sdp[r], (size_t)scn[r], kNcclDouble, r, c.nccl, st);
                if (rc2 == 0 && rcn[r]) rc2 = g_rccl.Recv(recv + rdp[r], (size_t)rcn[r], kNcclDouble, r, c.nccl, st);
            }
            int rc3 = g_rccl.GroupEnd();
            if (rc2 || rc3) return fail(EDGPU_ECOMM, std::string("edgpu: RCCL send/recv failed: ") + g_rccl.GetErrorString(rc2 ? rc2 : rc3));
        } else if (c.kind == Comm::CALLBACK) {
            if (!c.xch) return fail(EDGPU_ECOMM, "edgpu: the callback exchange needs an all-to-all-v callback");
            std::vector<int64_t> s2 = scn, r2 = rcn; s2[c.rank] = 0; r2[c.rank] = 0;       // own block already copied
            HIPCHK(hipStreamSynchronize(st));
            if (c.xch(c.ctx, send, s2.data(), sdp.data(), recv, r2.data(), rdp.data(), P)) return fail(EDGPU_ECOMM, "edgpu: exchange callback failed");
        } else if (c.kind == Comm::EMULATED) {
            for (int r = 0; r < P; r++)
                if (r != c.rank) CHK(emu_fill(recv + rdp[r], rcn[r], send + sdp[c.rank], scn[c.rank], st));
        } else
            return fail(EDGPU_ECOMM, "edgpu: mpi_size>1 but no communicator attached (edgpu_comm_init_*)");
        return EDGPU_OK;
    };
    const unsigned tgrid = (unsigned)(cdiv(H.Qdw, 32) * cdiv(H.DimUp, 32));
    hipLaunchKernelGGL(k_transpose2d, dim3(tgrid), dim3(256), 0, st, v, s->tA.p, H.Qdw, H.DimUp);     // tA: DimUp x Qdw
    HIPCHK(hipGetLastError());
    {
        // the chunked LDS pipeline below issues its own forward messages
        int64_t qmin = s->Qup; for (int r = 0; r < P; r++) { int64_t q, u; split(H.DimUp, r, q, u); qmin = std::min(qmin, q); }
        const bool chunked = s->tr_lds && std::min<int64_t>(s->tr_chunks, qmin) > 1;
        if (!chunked) CHK(a2a(s->tA.p, sc, sd, s->tB.p, rc, rd));                                     // tB: blocks by sender
    }
    if (s->tr_lds) {
        // rows fit LDS: read the received blocks and write the blocks of the way back directly (tVT is the out buffer).
        // Pipelined over C chunks of the up-block rows (as the reference sends its transposes column by column,
        // ED_HAMILTONIAN_COMMON.f90:96-113): chunk c of every rank's up-block is a contiguous piece of each message in both
        // directions, so the all-to-all of chunk c+1 is in flight on the comm stream while chunk c runs its row pass on a
        // stream of its own, and the way back of chunk c follows as soon as that pass has finished.  C = 1: the serial form.
        TdwFusedArgs F{}; F.DimDw = (int32_t)H.DimDw; F.dww = s->dwf->width; F.P = P; F.Qup = s->Qup;
        for (int r = 0; r <= P; r++) F.dw0[r] = Bk.dw0[r];
        F.ellc = s->dwf->ellc.p; F.ellv = s->dwf->ellv.p; F.blk_in = s->tB.p; F.blk_out = s->tVT.p;
#ifdef EDGPU_EXPERIMENTS
        if (s->dbg & 2) F.dww = 0;      // timing only: no dw hops (wrong results)
#endif
        static OncePerDevice once;
        if (once.first(s->device)) { if (hipFuncSetAttribute((const void *)k_tdw_blocks, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) (void)hipGetLastError(); }
        std::vector<int64_t> qu(P), u0(P), qd(P), d0(P);
        for (int r = 0; r < P; r++) { split(H.DimUp, r, qu[r], u0[r]); split(H.DimDw, r, qd[r], d0[r]); }
        int64_t qmin = s->Qup; for (int r = 0; r < P; r++) qmin = std::min(qmin, qu[r]);
        const int C = (int)std::max<int64_t>(1, std::min<int64_t>(s->tr_chunks, qmin));
        if (C == 1) {
            // forward exchange was issued above as one message per peer
            if (s->Qup > 0) hipLaunchKernelGGL(k_tdw_blocks, dim3((unsigned)s->Qup), dim3(H.DimDw >= 4096 ? 1024 : 256), (size_t)H.DimDw * 8, st, F);
            HIPCHK(hipGetLastError());
            CHK(a2a(s->tVT.p, rc, rd, s->tA.p, sc, sd));                                              // tA: DimUp x Qdw again
            return EDGPU_OK;
        }
        if (!s->tstream) CHK(g_streams.take(s->device, &s->tstream));
        while ((int)s->tr_ev.size() < 2 * C) { hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); s->tr_ev.push_back(e); }
        auto cut = [&](int64_t q, int c) { return q * c / C; };                                       // chunk c of a block of q rows: [cut(q,c), cut(q,c+1))
        std::vector<int64_t> sc2(P), sd2(P), rc2(P), rd2(P);
        auto counts = [&](int c) {
            for (int r = 0; r < P; r++) {
                const int64_t a = cut(qu[r], c), b = cut(qu[r], c + 1), ma = cut(s->Qup, c), mb = cut(s->Qup, c + 1);
                sc2[r] = (b - a) * H.Qdw; sd2[r] = H.Qdw * (u0[r] + a);                              // rows u0_r+a .. of my transposed shard -> rank r
                rc2[r] = (mb - ma) * qd[r]; rd2[r] = s->Qup * d0[r] + ma * qd[r];                      // my rows ma.. out of rank r's dw columns
            }
        };
        // comm stream: F(0) F(1) B(0) F(2) B(1) ... ; compute stream: row pass of chunk c after F(c)
        auto forward = [&](int c) -> int { counts(c); CHK(a2a(s->tA.p, sc2, sd2, s->tB.p, rc2, rd2)); HIPCHK(hipEventRecord(s->tr_ev[2 * c], st)); return EDGPU_OK; };
        auto rowpass = [&](int c) -> int {
            HIPCHK(hipStreamWaitEvent(s->tstream, s->tr_ev[2 * c], 0));
            const int64_t ma = cut(s->Qup, c), mb = cut(s->Qup, c + 1);
            TdwFusedArgs Fc = F; Fc.il0 = ma;
            if (mb > ma) hipLaunchKernelGGL(k_tdw_blocks, dim3((unsigned)(mb - ma)), dim3(H.DimDw >= 4096 ? 1024 : 256), (size_t)H.DimDw * 8, s->tstream, Fc);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(s->tr_ev[2 * c + 1], s->tstream));
            return EDGPU_OK;
        };
        auto backward = [&](int c) -> int { HIPCHK(hipStreamWaitEvent(st, s->tr_ev[2 * c + 1], 0)); counts(c); CHK(a2a(s->tVT.p, rc2, rd2, s->tA.p, sc2, sd2)); return EDGPU_OK; };
        for (int c = 0; c < C; c++) {
            CHK(forward(c));
            CHK(rowpass(c));
            if (c > 0) CHK(backward(c - 1));
        }
        CHK(backward(C - 1));
        return EDGPU_OK;
    }
    const unsigned bgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(s->Qup * H.DimDw, 256), 65536));
    Bk.blocks = s->tB.p; Bk.vt = s->tVT.p; Bk.dir = 0;
    hipLaunchKernelGGL(k_tblocks, dim3(bgrid), dim3(256), 0, st, Bk);
    TdwArgs D{}; D.DimDw = (int32_t)H.DimDw; D.dww = s->dwf->width; D.use_lds = 0;
    D.ellc = s->dwf->ellc.p; D.ellv = s->dwf->ellv.p; D.vt = s->tVT.p; D.hvt = s->tHVT.p;
#ifdef EDGPU_EXPERIMENTS
    if (s->dbg & 2) D.dww = 0;
#endif
    if (s->Qup > 0 && s->dwf->packed_elem8 && s->up_big && D.dww > 0) {
        // long rows: the packed-image gather kernel of the local pass, applied to H_dw (no diagonal)
        UpBigArgs U{};
        U.DimUp = (int32_t)H.DimDw; U.DimPh = 1; U.rsplit = s->row_split; U.Qdw = s->Qup; U.dw0 = 0; U.nrows = s->Qup;
        U.packed8 = reinterpret_cast<const uint4 *>(s->dwf->pelem8.p); U.tval = s->dwf->ptval.p; U.nterms = s->dwf->pterms;
        U.v = s->tVT.p; U.out = s->tHVT.p; U.nodiag = 1;
        hipLaunchKernelGGL((k_uppass_big<false, false>), dim3((unsigned)(cdiv(U.nrows, 8) * 8 * s->row_split)), dim3(512), 0, st, U);
    } else if (s->Qup > 0) hipLaunchKernelGGL(k_tdw, dim3((unsigned)s->Qup), dim3(H.DimDw >= 4096 ? 1024 : 256), 0, st, D);
    Bk.vt = s->tHVT.p; Bk.dir = 1;
    hipLaunchKernelGGL(k_tblocks, dim3(bgrid), dim3(256), 0, st, Bk);                                  // tB: blocks by destination
    HIPCHK(hipGetLastError());
    CHK(a2a(s->tB.p, rc, rd, s->tA.p, sc, sd));                                                       // tA: DimUp x Qdw again
    return EDGPU_OK;
}

// ---------------------------------------------------------------------------------------
// fused Lanczos iteration (single rank, two-factor operator of a large sector, stored or on the fly): see k_uppass2_ax
// ---------------------------------------------------------------------------------------
#ifndef LZAX_NT
#define LZAX_NT 1024
#endif
// Can hxv() of this sector take vectors with a padded row pitch?  Only the two-pass product of a single rank whose local
// pass is one of the LDS-staged kernels and whose dw pass is the vectorised column-tiled one (Jx/Jp through k_ndpass).
// Odd DimUp: the vectorised dw pass and the fused iteration work on column PAIRS.  With the padded row pitch every row ends in at
// least one zero column, so the pass simply runs over DimUp + 1 columns (the padding column stays zero: its hops read zeros, its
// local part is never written).  All sectors of Ns = 7 (the reference's default Nbath = 6) and Ns = 15 have odd DimUp.
static inline bool odd_cols_ok(const S *s) { return (s->H.DimUp % 2 == 0) || s->pad_allowed; }
static inline int32_t dw_cols(const S *s, int64_t ld) { return (int32_t)((s->H.DimUp % 2 != 0 && ld > s->H.DimUp) ? s->H.DimUp + 1 : s->H.DimUp); }
static bool pitch_ok(const S *s)
{
    const SectorHost &H = s->H;
    if (H.desc.storage == EDGPU_STORE_CSR) return s->csr_ld > H.DimUp;      // explicit CSR whose columns index the pitched space (single rank)
    if (s->multi || !s->kron2 || !s->kron_split || H.DimPh != 1 || s->tr_on) return false;
    if (s->extra && !(s->has_nd && !s->has_ph && s->nd_split)) return false;
    if (!(s->dw_variant >= 2 && odd_cols_ok(s) && s->dw_w >= 4 && s->dwf->width <= 64)) return false;
    const bool persistent = s->up_persistent && s->upf->packed && s->upf->width <= 16 && s->use_lds && H.DimUp <= 13312;
    const bool halo = !s->use_lds && s->upf->seg_ok && s->up_halo && !s->extra;
    return persistent || halo;
}
// which local-half kernel the fused iteration uses: 2 = pair form (<= 8 hops per up configuration), 3 = pair form with 10 hop
// slots (9-10 hops: one or two electrons off half filling), 1 = one configuration per lane (<= 16 hops), 0 = none (rows beyond
// LDS, wider factors)
static int fused_form(const S *s)
{
    if (!s->upf || !s->up_persistent || !s->use_lds || s->H.DimUp > 13312) return 0;
    if (s->up_pairs && s->upf->packed_elem && s->upf->width <= 8 && odd_cols_ok(s)) return 2;
    if (s->up_pairs && s->up_pairs10 && s->upf->packed_elem8 && s->upf->width <= 10 && odd_cols_ok(s)) return 3;      // pair form with 10 hop slots (8-dword element image)
    return (s->upf->packed && s->upf->width <= 16) ? 1 : 0;
}
static bool fused_ok(const S *s)
{
    const SectorHost &H = s->H;
    // Jx/Jp sectors run their non-local pass between the two halves; phonon terms live in the row-wise kernel only
    const bool terms_ok = !s->extra || (s->has_nd && !s->has_ph && s->nd_split);
    return s->lz_fused_on && !s->multi && s->kron2 && terms_ok && s->kron_split && H.desc.storage != EDGPU_STORE_CSR && fused_form(s) != 0 &&
           odd_cols_ok(s) && s->dw_variant >= 2 && s->dw_w >= 4 && s->dwf->width <= 64;
}
// local half: r <- u_k = r - alpha s u_prev (in place), TMP = |u_k|^2 and the norm step, t <- local part of H u_k (unscaled)
// redo: r already IS u_k (it was changed in place after the pass that formed it): alpha is zeroed so that r stays as it is,
// and the scalar step that follows the norm does not rotate the scales again (k_scalar_step modes 9 / 8)
static int fused_local(S *s, double *r, const double *uprev, double *t, double *acc, bool redo = false)
{
    if (redo) { hipLaunchKernelGGL(k_scalar_step, dim3(1), dim3(1), 0, s->stream, s->scal.p, s->alphas.p, s->betas.p, (const double *)nullptr, -1, 9); HIPCHK(hipGetLastError()); }
    const SectorHost &H = s->H; const edgpu_desc &d = H.desc;
    const bool direct = d.storage == EDGPU_STORE_DIRECT;
    if (direct) CHK(direct_tables(s));
    UpAxArgs U{};
    U.DimUp = (int32_t)H.DimUp; U.DimPh = H.DimPh; U.Qdw = H.Qdw; U.dw0 = H.dw0; U.nrows = H.Qdw * H.DimPh;
    if (fused_form(s) >= 2 && (H.DimUp & 1)) { U.DimUp = (int32_t)H.DimUp + 1; U.dtrue = (int32_t)H.DimUp; }      // odd rows, pair forms: the last pair ends in the padding column of the pitched vectors
    U.packed = reinterpret_cast<const uint4 *>(fused_form(s) == 3 ? s->upf->pelem8.p : s->upf->pelem.p); U.tval = s->upf->ptval.p; U.nterms = s->upf->pterms;
    U.diag = s->diag.p; U.Eup = s->upf->E.p; U.Edw = s->dwf->E.p; U.cross = s->cross.p; U.mapup = s->upf->map.p; U.mapdw = s->dwf->map.p;
    U.impmask = (1 << d.Norb) - 1; U.nI = 1 << d.Norb; U.econst = H.econst;
    U.r = r; U.vprev = uprev; U.t_out = t; U.scal = s->scal.p; U.partial = s->partial.p; U.acc = acc; U.zvec = s->zvec.p; U.ld = (int32_t)s->vld;
    const size_t lds = (((size_t)H.DimUp + 1) & ~(size_t)1) * 8 + 128 * 8 + 32 * 8 + 32 * 8 + (direct ? (((size_t)H.DimUp + 16) & ~(size_t)15) : 0);      // + the impurity-word bytes of the on-the-fly form
    const int grid = (int)std::min<int64_t>(U.nrows, s->up_grid);
    U.pcols = s->upf->pcols.p; U.pcodes = s->upf->pcodes.p; U.upw = s->upf->width; U.crowtab = direct ? s->crow_tab.p : nullptr;
    U.pelem16 = (s->upf->packed_elem16 && s->up_elem16) ? reinterpret_cast<const uint4 *>(s->upf->pelem16.p) : nullptr;
    if (direct && !U.crowtab) return fail(EDGPU_EINVAL, "edgpu: internal: row-constant table of the on-the-fly form missing");
    const dim3 g(grid), b(LZAX_NT);
#define AX_LAUNCH(K_, D_, A_)                                                                                                   \
    do {                                                                                                                            \
        static OncePerDevice once_;                                                                                                 \
        if (once_.first(s->device)) (void)hipFuncSetAttribute((const void *)K_<D_, A_, LZAX_NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((K_<D_, A_, LZAX_NT>), g, b, lds, s->stream, U);                                                         \
    } while (0)
#define AX10_LAUNCH(D_, A_)                                                                                                     \
    do {                                                                                                                            \
        static OncePerDevice once_;                                                                                                 \
        if (once_.first(s->device)) (void)hipFuncSetAttribute((const void *)k_uppass2_ax<D_, A_, LZAX_NT, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((k_uppass2_ax<D_, A_, LZAX_NT, 10>), g, b, lds, s->stream, U);                                           \
    } while (0)
    if (fused_form(s) == 3) {
        if (direct) { if (acc) AX10_LAUNCH(true, true); else AX10_LAUNCH(true, false); }
        else { if (acc) AX10_LAUNCH(false, true); else AX10_LAUNCH(false, false); }
    } else
    if (fused_form(s) == 2) {
        if (direct) { if (acc) AX_LAUNCH(k_uppass2_ax, true, true); else AX_LAUNCH(k_uppass2_ax, true, false); }
        else { if (acc) AX_LAUNCH(k_uppass2_ax, false, true); else AX_LAUNCH(k_uppass2_ax, false, false); }
    } else {
        if (direct) { if (acc) AX_LAUNCH(k_uppass_ax, true, true); else AX_LAUNCH(k_uppass_ax, true, false); }
        else { if (acc) AX_LAUNCH(k_uppass_ax, false, true); else AX_LAUNCH(k_uppass_ax, false, false); }
    }
#undef AX_LAUNCH
#undef AX10_LAUNCH
    HIPCHK(hipGetLastError());
    s->fuse_mode = redo ? 8 : 4; s->fuse_zvec = nullptr;
    const int rc = reduce_partials(s, grid);
    s->fuse_mode = -1;
    if (rc != EDGPU_OK) return rc;
    if (s->has_nd) {       // spin-exchange / pair-hopping terms of u_k (complete now), unscaled, into the local part
        NdArgs Nd{};
        Nd.DimUp = (int32_t)H.DimUp; Nd.DimDw = (int32_t)H.DimDw; Nd.DimPh = H.DimPh; Nd.Norb = d.Norb; Nd.Qdw = H.Qdw; Nd.dw0 = H.dw0;
        Nd.mapup = s->upf->map.p; Nd.mapdw = s->dwf->map.p; Nd.rankup = s->upf->rank.p; Nd.rankdw = s->dwf->rank.p; Nd.Jx = d.Jx; Nd.Jp = d.Jp;
        Nd.vfull = r; Nd.out = t; Nd.scal = s->scal.p; Nd.xmap = nullptr; Nd.ld = (int32_t)s->vld;
        hipLaunchKernelGGL(k_ndpass<false>, dim3((unsigned)(H.Qdw * H.DimPh)), dim3(256), 0, s->stream, Nd);
        HIPCHK(hipGetLastError());
    }
    return EDGPU_OK;
}
// dw half: t <- r_k = s_k (t + dw hops of u) - beta_{k-1} s_{k-1} u_prev ; TMP = (s_k u).r_k and the alpha step
static int fused_dw(S *s, const double *u, double *t, const double *uprev)
{
    const SectorHost &H = s->H;
    DwArgs D{};
    D.DimUp = dw_cols(s, s->vld); D.DimDw = (int32_t)H.DimDw; D.DimPh = H.DimPh; D.Qdw = H.Qdw; D.dw0 = H.dw0;
    D.dwc = s->dwf->ellc.p; D.dwv = s->dwf->ellv.p; D.dww = s->dwf->width; D.v = u; D.vfull = u; D.out = t; D.scal = s->scal.p; D.partial = s->partial.p;
    D.uprev2 = uprev;
    D.ld = (int32_t)s->vld;
    D.w = s->dw_w; D.wshift = 0; while ((1 << D.wshift) < D.w) D.wshift++;
    D.ntiles = (int32_t)cdiv(D.DimUp, D.w);
    D.R = std::min(64, s->dw_rows);
    D.chunks = (int32_t)cdiv(H.Qdw * H.DimPh, D.R);
    const int64_t nblocks = (int64_t)D.ntiles * D.chunks;
    const bool nt = s->dw_variant != 2, rag = s->dwf->ragged;
    const size_t dlds = std::max<size_t>((size_t)D.R * D.dww * 12 + 16, (size_t)s->dw_lds_kb * 1024);
#define DW2_LAUNCH(N_, R_)                                                                                                      \
    do {                                                                                                                            \
        static OncePerDevice once_;                                                                                                 \
        if (dlds > 60 * 1024 && once_.first(s->device)) {                                                                           \
            if (hipFuncSetAttribute((const void *)k_dwpass2<true, N_, R_>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) (void)hipGetLastError(); \
        }                                                                                                                           \
        hipLaunchKernelGGL((k_dwpass2<true, N_, R_>), dim3((unsigned)nblocks), dim3(256), dlds, s->stream, D);                      \
    } while (0)
    if (nt) { if (rag) DW2_LAUNCH(true, true); else DW2_LAUNCH(true, false); } else { if (rag) DW2_LAUNCH(false, true); else DW2_LAUNCH(false, false); }
#undef DW2_LAUNCH
    HIPCHK(hipGetLastError());
    s->fuse_mode = 5; s->fuse_zvec = nullptr;
    const int rc = reduce_partials(s, nblocks);
    s->fuse_mode = -1;
    return rc;
}

// `ld`: row pitch of v / out / uprev in doubles when the Lanczos loops run on pitched vectors (pitch_ok sectors only), 0 = dense
static int hxv(S *s, const double *v, double *out, bool epi, const double *uprev, int64_t ld = 0)
{
    if (ld == s->H.DimUp) ld = 0;
    if (ld && !pitch_ok(s)) return fail(EDGPU_EINVAL, "edgpu: internal: pitched vectors on a product form without pitch support");
    const SectorHost &H = s->H; const edgpu_desc &d = H.desc;
    const bool multi = s->multi;
    const bool direct = d.storage == EDGPU_STORE_DIRECT;
    if (direct) CHK(direct_tables(s));
    // explicit CSR with columns stored in the pitched space: a caller's dense vectors (plain products only) go through pitched
    // scratch copies -- made before the exchange starts, which reads the pitched copy
    const bool via_scratch = d.storage == EDGPU_STORE_CSR && s->csr_ld > H.DimUp && ld == 0;
    double *dense_out = out;
    if (via_scratch) {
        if (epi) return fail(EDGPU_EINVAL, "edgpu: internal: Lanczos epilogue on dense vectors with pitched CSR columns");
        const size_t rows = (size_t)(s->nloc / H.DimUp), vl = rows * (size_t)s->csr_ld;
        if (!s->csr_xs.p) { CHK(s->csr_xs.alloc(vl)); CHK(s->csr_ys.alloc(vl)); HIPCHK(hipMemsetAsync(s->csr_xs.p, 0, vl * 8, s->stream)); HIPCHK(hipMemsetAsync(s->csr_ys.p, 0, vl * 8, s->stream)); }
        HIPCHK(hipMemcpy2DAsync(s->csr_xs.p, (size_t)s->csr_ld * 8, v, (size_t)H.DimUp * 8, (size_t)H.DimUp * 8, rows, hipMemcpyDeviceToDevice, s->stream));
        v = s->csr_xs.p; out = s->csr_ys.p;
    }
    if (multi) {  // start the exchange on the comm stream; local work overlaps it
        s->packed_ahead = false;
        if (s->halo && !s->tr_on) { CHK(halo_pack(s, v, s->stream)); s->packed_ahead = true; }      // the pack kernel first, alone, on the compute stream
        HIPCHK(hipEventRecord(s->ev_ready, s->stream));
        HIPCHK(hipStreamWaitEvent(s->cstream, s->ev_ready, 0));
        const bool timed = s->comm_slot >= 0 && 2 * s->comm_slot + 1 < (int)s->comm_ev.size();
        if (timed) HIPCHK(hipEventRecord(s->comm_ev[2 * s->comm_slot], s->cstream));
        if (s->tr_on) CHK(transposed_dw_pass(s, v, s->cstream));      // whole dw part, on the comm stream
        else CHK(gather_vector(s, v, s->cstream));
        s->packed_ahead = false;
        if (timed) HIPCHK(hipEventRecord(s->comm_ev[2 * s->comm_slot + 1], s->cstream));
        HIPCHK(hipEventRecord(s->ev_gather, s->cstream));
    }
    int64_t nblocks = 0;
    if (d.storage == EDGPU_STORE_CSR) {
        // split CSR: the local block only reads columns of the rank's own dw columns, so it runs on the
        // shard itself (pointer shifted to global column numbering) while the exchange is in flight
        const bool overlap = multi && s->csr_split && H.DimPh == 1;
        if (multi && !overlap) HIPCHK(hipStreamWaitEvent(s->stream, s->ev_gather, 0));
        CsrArgs A{};
        A.ld = s->csr_ld;
        A.nrows = s->nloc_el; A.rows_per_block = s->csr_rpb; A.DimUp = (int32_t)H.DimUp; A.blocks_per_row = s->csr_bpr; A.Qdw = H.Qdw; A.tile = s->csr_tile;
        A.rowptr = s->csr_rowptr.p; A.cols = s->csr_cols.p; A.vals = s->csr_vals.p;
        A.x = overlap ? (s->halo ? v : v - H.dw0 * H.DimUp) : (multi ? s->vfull.p : v); A.y = out; A.v_local = v; A.uprev = uprev; A.scal = s->scal.p; A.partial = s->partial.p; A.epi = epi; A.xcd_rows = s->csr_xcd_rows;
        // the column index is global; on one GPU local == global
        // Phonon sectors: the electronic blocks are the same for every phonon block iph; they run once per iph on the
        // slices x(:,iph), y(:,iph) (x-space of one block: Xel elements; phonon sectors of several ranks use the full gather)
        const int nph = H.DimPh;
        const int64_t Xel = H.DimUp * (multi ? s->nhalo : H.DimDw);
        auto launch = [&](const CsrArgs &B0, bool small, bool nt, int64_t nb, bool il = false) {
            for (int iph = 0; iph < nph; iph++) {
                CsrArgs B = B0;
                B.x += (int64_t)iph * Xel; B.y += (int64_t)iph * s->nloc_el; B.v_local += (int64_t)iph * s->nloc_el;
                if (B.uprev) B.uprev += (int64_t)iph * s->nloc_el;
                B.partial += (int64_t)iph * nb;
                if (il) {
                    if (B.coltiles) { if (nt) hipLaunchKernelGGL((k_csr_il<true, 1>), dim3((unsigned)nb), dim3(64), 0, s->stream, B); else hipLaunchKernelGGL((k_csr_il<false, 1>), dim3((unsigned)nb), dim3(64), 0, s->stream, B); }
                    else { if (nt) hipLaunchKernelGGL((k_csr_il<true, 0>), dim3((unsigned)nb), dim3(64), 0, s->stream, B); else hipLaunchKernelGGL((k_csr_il<false, 0>), dim3((unsigned)nb), dim3(64), 0, s->stream, B); }
                }
                else if (small) { if (nt) hipLaunchKernelGGL((k_csr_spmv2<64, 640, true>), dim3((unsigned)nb), dim3(64), 0, s->stream, B); else hipLaunchKernelGGL((k_csr_spmv2<64, 640, false>), dim3((unsigned)nb), dim3(64), 0, s->stream, B); }
                else { if (nt) hipLaunchKernelGGL((k_csr_spmv2<64, 1152, true>), dim3((unsigned)nb), dim3(64), 0, s->stream, B); else hipLaunchKernelGGL((k_csr_spmv2<64, 1152, false>), dim3((unsigned)nb), dim3(64), 0, s->stream, B); }
            }
        };
        nblocks = (int64_t)s->csr_bpr * H.Qdw;
        A.pass = s->csr_split ? 1 : 0; A.coltiles = 0;
        launch(A, s->csr_split && s->csrL_small, s->csr_il_loc && s->csrL_nt, nblocks, s->csr_split && s->csr_il_loc);
        HIPCHK(hipGetLastError());
        if (s->csr_split) {
            A.rows_per_block = s->csrD_rpb; A.blocks_per_row = s->csrD_bpr; A.coltiles = 1; A.xcd_rows = 0;
            nblocks = (int64_t)s->csrD_bpr * H.Qdw;
            if (s->csr_three) {
                // dw hops into the rank's own rows: their columns are the first Qdw rows of the x-space, i.e. the local
                // shard itself -- this block too runs while the exchange is in flight
                A.rowptr = s->csrO_rowptr.p; A.cols = s->csrO_cols.p; A.vals = s->csrO_vals.p; A.pass = 3;
                if (s->csr_uni) { A.rowbase = s->csrO_base.p; A.rowcnt = s->csrO_cnt.p; A.cpre = s->csr_tm ? s->csrO_cpre.p : nullptr; }
                launch(A, s->csrD_small, s->csr_il_dw && s->csrD_nt != 0, nblocks, s->csr_il_dw);
                HIPCHK(hipGetLastError());
            }
            if (overlap) { HIPCHK(hipStreamWaitEvent(s->stream, s->ev_gather, 0)); A.x = s->vfull.p; }
            if (s->nd_nr > 0) {     // stored non-local block (Jx/Jp): compressed rows, wave-level segmented sums
                for (int iph = 0; iph < nph; iph++) {
                    NdCsrArgs Nd{};
                    Nd.nr = s->nd_nr; Nd.rows = s->nd_rows.p; Nd.ptr = s->nd_ptr.p; Nd.cols = s->nd_cols.p; Nd.vals = s->nd_vals.p; Nd.slot = s->nd_slot.p;
                    Nd.x = (multi ? s->vfull.p : v) + (int64_t)iph * Xel; Nd.y = out + (int64_t)iph * s->nloc_el; Nd.scal = s->scal.p;
                    const unsigned g = (unsigned)cdiv(s->nd_nr, 256);
                    if (epi) hipLaunchKernelGGL(k_csr_nd<true>, dim3(g), dim3(256), 0, s->stream, Nd);
                    else hipLaunchKernelGGL(k_csr_nd<false>, dim3(g), dim3(256), 0, s->stream, Nd);
                }
                HIPCHK(hipGetLastError());
            }
            if (nph > 1) {      // stored phonon part: w0 n_ph and g(iel) (b + b^+); phonon neighbours are rows of this rank
                const unsigned g = (unsigned)std::min<int64_t>(cdiv(s->nloc, 256), 65536);
                if (epi) hipLaunchKernelGGL(k_csr_ph<true>, dim3(g), dim3(256), 0, s->stream, s->nloc_el, (int32_t)nph, d.w0_ph, (const double *)s->csr_gel.p, v, out, (const double *)s->scal.p);
                else hipLaunchKernelGGL(k_csr_ph<false>, dim3(g), dim3(256), 0, s->stream, s->nloc_el, (int32_t)nph, d.w0_ph, (const double *)s->csr_gel.p, v, out, (const double *)s->scal.p);
                HIPCHK(hipGetLastError());
            }
            A.rowptr = s->csrD_rowptr.p; A.cols = s->csrD_cols.p; A.vals = s->csrD_vals.p;
            if (s->csr_uni) { A.rowbase = s->csrD_base.p; A.rowcnt = s->csrD_cnt.p; A.cpre = s->csr_tm ? s->csrD_cpre.p : nullptr; }
            A.pass = 2;
            nblocks = (int64_t)s->csrD_bpr * H.Qdw;
            launch(A, s->csrD_small, s->csrD_nt != 0, nblocks, s->csr_il_dw);
            HIPCHK(hipGetLastError());
        }
        nblocks *= nph;      // one partial per workgroup and phonon block
        if (via_scratch) HIPCHK(hipMemcpy2DAsync(dense_out, (size_t)H.DimUp * 8, out, (size_t)s->csr_ld * 8, (size_t)H.DimUp * 8, (size_t)(s->nloc / H.DimUp), hipMemcpyDeviceToDevice, s->stream));
    } else if (s->kron2) {
        Kron2Args A{};
        A.DimUp = (int32_t)H.DimUp; A.DimDw = (int32_t)(s->halo ? s->nhalo : H.DimDw); A.DimPh = H.DimPh; A.Norb = d.Norb; A.Qdw = H.Qdw; A.dw0 = H.dw0;
        A.upc = s->upf->ellc.p; A.upv = s->upf->ellv.p; A.upw = s->upf->width;
        A.dwc = s->halo ? s->dw_ellc_x.p : s->dwf->ellc.p; A.dwv = s->dwf->ellv.p; A.dww = s->dwf->width;
        A.diag = s->diag.p; A.Eup = s->upf->E.p; A.Edw = s->dwf->E.p; A.cross = s->cross.p;
        A.mapup = s->upf->map.p; A.mapdw = s->dwf->map.p; A.rankup = s->upf->rank.p; A.rankdw = s->dwf->rank.p;
        A.impmask = (1 << d.Norb) - 1; A.nI = 1 << d.Norb; A.econst = H.econst; A.w0 = d.w0_ph; A.gsum = H.gph_sum; A.Jx = d.Jx; A.Jp = d.Jp;
        A.gup = s->gup.p; A.gdw = s->gdw.p;
        A.v = v; A.vfull = multi ? s->vfull.p : v; A.out = out; A.uprev = uprev; A.scal = s->scal.p; A.partial = s->partial.p;
        A.sr_part = s->sr_pending ? s->partial_b.p : nullptr; A.sr_n = s->sr_pending ? s->sr_nb : 0; A.scal_w = s->scal.p; A.betas_w = s->betas.p;
#ifdef EDGPU_EXPERIMENTS
        if (s->dbg & 1) A.upw = 0;      // timing only: no up hops / no dw hops (wrong results)
        if (s->dbg & 2) A.dww = 0;
#endif
        int base = (direct ? KF_DIRECT : 0) | (epi ? KF_EPI : 0) | (s->extra ? (KF_PH | KF_ND) : 0);
        nblocks = H.Qdw * H.DimPh;
        // phonon terms live in the fused kernel only; Jx/Jp (without phonons) have their own pass
        const bool nd_pass = s->has_nd && !s->has_ph && s->nd_split;
        const bool split = multi || (s->kron_split && (!s->extra || nd_pass));
        if (!split) CHK(dispatch_kron2(s, base | (s->use_lds ? KF_LDS : 0) | ((epi && s->sr_now) ? KF_SR : 0), A, nblocks));
        else {
            const bool plain = !s->extra || nd_pass;      // local pass = stored/on-the-fly diagonal + up hops only
            // k_uppass_halo writes the unscaled local part; under the Lanczos epilogue the vectorised dw pass finishes it (uprev2)
            const bool dw_vec2 = s->dw_variant >= 2 && (H.DimUp % 2 == 0 || ld > H.DimUp) && s->dw_w >= 4 && s->dwf->width <= 64;      // odd DimUp: pitched vectors only (dw_cols)
            const bool halo_can = !s->use_lds && plain && H.DimPh == 1 && s->upf->seg_ok && s->up_halo && (!epi || (dw_vec2 && !s->extra && !s->tr_on));
            bool halo_epi = false;
            const bool persistent = s->up_persistent && plain && s->upf->packed && s->upf->width <= 16 && s->use_lds && H.DimUp <= 13312;
            // the column-tiled dw pass
            auto run_dw = [&](bool dw_epi, const double *uprev2) -> int {
                DwArgs D{};
                D.DimUp = A.DimUp; D.DimDw = A.DimDw; D.DimPh = A.DimPh; D.Qdw = A.Qdw; D.dw0 = A.dw0;
                D.dwc = A.dwc; D.dwv = A.dwv; D.dww = A.dww; D.v = v; D.vfull = A.vfull; D.out = out; D.scal = s->scal.p; D.partial = s->partial.p;
                D.uprev2 = uprev2; D.ld = (int32_t)ld; D.own_in_v = (multi && s->x_own_in_v) ? 1 : 0;
#ifdef EDGPU_EXPERIMENTS
                D.abl = XKNOB("EDGPU_DW_ABL", 0);
#endif
                D.w = s->dw_w; D.wshift = 0; while ((1 << D.wshift) < D.w) D.wshift++;
                const int dv = s->dw_variant;
                const bool vec2 = dv >= 2 && (H.DimUp % 2 == 0 || ld > H.DimUp) && D.w >= 4 && D.dww <= 64;
                if (vec2) D.DimUp = dw_cols(s, ld);
                D.ntiles = (int32_t)cdiv(D.DimUp, D.w);
                D.R = vec2 ? std::min(64, s->dw_rows) : 256 / D.w;
                D.chunks = (int32_t)cdiv(H.Qdw * H.DimPh, D.R);
                nblocks = (int64_t)D.ntiles * D.chunks;
                if (!vec2) {
                    if (dw_epi) hipLaunchKernelGGL(k_dwpass<true>, dim3((unsigned)nblocks), dim3(256), 0, s->stream, D);
                    else hipLaunchKernelGGL(k_dwpass<false>, dim3((unsigned)nblocks), dim3(256), 0, s->stream, D);
                } else {
                    const bool nt = dv != 2, rag = s->dwf->ragged;
                    // extra LDS (EDGPU_DW_LDS_KB) caps the workgroups per CU; with 16-row chunks full residency measured best (Ns=16 and Ns=18 sweeps, round 2)
                    const size_t dlds = std::max<size_t>((size_t)D.R * D.dww * 12 + 16, (size_t)s->dw_lds_kb * 1024);
#define DW_LAUNCH(E_, N_, R_)                                                                                                   \
    do {                                                                                                                            \
        static OncePerDevice once_;                                                                                                 \
        if (dlds > 60 * 1024 && once_.first(s->device)) {      /* static LDS of the kernel comes on top: stay below the 160 KB of a CU */ \
            if (hipFuncSetAttribute((const void *)k_dwpass2<E_, N_, R_>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) (void)hipGetLastError(); \
        }                                                                                                                           \
        hipLaunchKernelGGL((k_dwpass2<E_, N_, R_>), dim3((unsigned)nblocks), dim3(256), dlds, s->stream, D);                        \
    } while (0)
                    if (dw_epi) { if (nt) { if (rag) DW_LAUNCH(true, true, true); else DW_LAUNCH(true, true, false); } else { if (rag) DW_LAUNCH(true, false, true); else DW_LAUNCH(true, false, false); } }
                    else { if (nt) { if (rag) DW_LAUNCH(false, true, true); else DW_LAUNCH(false, true, false); } else { if (rag) DW_LAUNCH(false, false, true); else DW_LAUNCH(false, false, false); } }
#undef DW_LAUNCH
                }
                HIPCHK(hipGetLastError());
                return EDGPU_OK;
            };
            const bool rows_pair = (H.DimUp % 2 == 0) || (ld > H.DimUp && s->pad_allowed);      // odd rows: pair kernels on pitched vectors only (the last pair ends in the padding column)
            const int32_t up_cols = (int32_t)(H.DimUp + (H.DimUp & 1)), up_dtrue = (H.DimUp & 1) ? (int32_t)H.DimUp : 0;
            const bool pairs10 = persistent && rows_pair && s->up_pairs && s->up_pairs10 && !(s->upf->packed_elem && s->upf->width <= 8) && s->upf->packed_elem8 && s->upf->width <= 10;
            if (pairs10) {      // pair form with 10 hop slots: the 8-dword element image (sectors one or two electrons off half filling)
                UpArgs2 U{};
                U.DimUp = up_cols; U.dtrue = up_dtrue; U.DimPh = A.DimPh; U.Qdw = A.Qdw; U.dw0 = A.dw0; U.nrows = H.Qdw * H.DimPh;
                U.packed = reinterpret_cast<const uint4 *>(s->upf->pelem8.p); U.tval = s->upf->ptval.p; U.nterms = s->upf->pterms;
                U.diag = A.diag; U.Eup = A.Eup; U.Edw = A.Edw; U.cross = A.cross; U.mapup = A.mapup; U.mapdw = A.mapdw; U.impmask = A.impmask; U.nI = A.nI; U.econst = A.econst;
                U.v = v; U.out = out; U.uprev = uprev; U.scal = s->scal.p; U.ld = (int32_t)ld;
                U.crowtab = direct ? s->crow_tab.p : nullptr;
                if (direct && !U.crowtab) return fail(EDGPU_EINVAL, "edgpu: internal: row-constant table of the on-the-fly form missing");
                const size_t lds = (size_t)up_cols * 8 + 128 * 8 + (direct ? 32 * 8 + (((size_t)up_cols + 15) & ~(size_t)15) : 0);
                const int grid = (int)std::min<int64_t>(U.nrows, s->up_grid);
#define UP10_LAUNCH(D_, E_)                                                                                                     \
    do {                                                                                                                            \
        static OncePerDevice once_;                                                                                                 \
        if (once_.first(s->device)) (void)hipFuncSetAttribute((const void *)k_uppass2<D_, E_, 1024, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((k_uppass2<D_, E_, 1024, 10>), dim3(grid), dim3(1024), lds, s->stream, U);                               \
    } while (0)
                if (direct) { if (epi) UP10_LAUNCH(true, true); else UP10_LAUNCH(true, false); }
                else { if (epi) UP10_LAUNCH(false, true); else UP10_LAUNCH(false, false); }
#undef UP10_LAUNCH
                HIPCHK(hipGetLastError());
            } else
            if (persistent && rows_pair && s->up_pairs && s->upf->packed_elem && s->upf->width <= 8) {
                UpArgs2 U{};
                U.DimUp = up_cols; U.dtrue = up_dtrue; U.DimPh = A.DimPh; U.Qdw = A.Qdw; U.dw0 = A.dw0; U.nrows = H.Qdw * H.DimPh;
                U.packed = reinterpret_cast<const uint4 *>(s->upf->pelem.p); U.tval = s->upf->ptval.p; U.nterms = s->upf->pterms;
                U.diag = A.diag; U.Eup = A.Eup; U.Edw = A.Edw; U.cross = A.cross; U.mapup = A.mapup; U.mapdw = A.mapdw; U.impmask = A.impmask; U.nI = A.nI; U.econst = A.econst;
                U.v = v; U.out = out; U.uprev = uprev; U.scal = s->scal.p; U.ld = (int32_t)ld;
                U.crowtab = direct ? s->crow_tab.p : nullptr;
                if (direct && !U.crowtab) return fail(EDGPU_EINVAL, "edgpu: internal: row-constant table of the on-the-fly form missing");
                const size_t lds = (size_t)up_cols * 8 + 128 * 8 + (direct ? 32 * 8 + (((size_t)up_cols + 15) & ~(size_t)15) : 0);
                const int grid = (int)std::min<int64_t>(U.nrows, s->up_grid);
                static OncePerDevice once2;
                if (once2.first(s->device)) {
                    (void)hipFuncSetAttribute((const void *)k_uppass2<false, false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    (void)hipFuncSetAttribute((const void *)k_uppass2<false, true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                }
                if (direct) {
                    static OncePerDevice once3;
                    if (once3.first(s->device)) {
                        (void)hipFuncSetAttribute((const void *)k_uppass2<true, false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                        (void)hipFuncSetAttribute((const void *)k_uppass2<true, true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    }
                    if (epi) hipLaunchKernelGGL((k_uppass2<true, true, 1024>), dim3(grid), dim3(1024), lds, s->stream, U);
                    else hipLaunchKernelGGL((k_uppass2<true, false, 1024>), dim3(grid), dim3(1024), lds, s->stream, U);
                } else if (epi) hipLaunchKernelGGL((k_uppass2<false, true, 1024>), dim3(grid), dim3(1024), lds, s->stream, U);
                else hipLaunchKernelGGL((k_uppass2<false, false, 1024>), dim3(grid), dim3(1024), lds, s->stream, U);
                HIPCHK(hipGetLastError());
            } else if (persistent) {
                UpArgs U{};
                U.DimUp = A.DimUp; U.DimPh = A.DimPh; U.Qdw = A.Qdw; U.dw0 = A.dw0; U.nrows = H.Qdw * H.DimPh;
                U.pcols = s->upf->pcols.p; U.pcodes = s->upf->pcodes.p; U.upw = s->upf->width; U.tval = s->upf->ptval.p; U.nterms = s->upf->pterms;
                U.pelem16 = (s->upf->packed_elem16 && s->up_elem16) ? reinterpret_cast<const uint4 *>(s->upf->pelem16.p) : nullptr;
                U.diag = A.diag; U.Eup = A.Eup; U.Edw = A.Edw; U.cross = A.cross; U.mapup = A.mapup; U.mapdw = A.mapdw; U.impmask = A.impmask; U.nI = A.nI; U.econst = A.econst;
                U.v = v; U.out = out; U.uprev = uprev; U.scal = s->scal.p; U.ld = (int32_t)ld;
                U.crowtab = direct ? s->crow_tab.p : nullptr;
                if (direct && !U.crowtab) return fail(EDGPU_EINVAL, "edgpu: internal: row-constant table of the on-the-fly form missing");
                const size_t lds = (((size_t)H.DimUp + 1) & ~(size_t)1) * 8 + 128 * 8 + (direct ? 32 * 8 + (((size_t)H.DimUp + 15) & ~(size_t)15) : 0);
                const int grid = (int)std::min<int64_t>(U.nrows, s->up_grid);
#define UP_LAUNCH(D_, E_, T_)                                                                                                   \
    do {                                                                                                                            \
        static OncePerDevice once_;                                                                                                 \
        if (once_.first(s->device)) (void)hipFuncSetAttribute((const void *)k_uppass<D_, E_, T_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((k_uppass<D_, E_, T_>), dim3(grid), dim3(T_), lds, s->stream, U);                                        \
    } while (0)
                if (s->up_threads == 512) {
                    if (direct) { if (epi) UP_LAUNCH(true, true, 512); else UP_LAUNCH(true, false, 512); }
                    else { if (epi) UP_LAUNCH(false, true, 512); else UP_LAUNCH(false, false, 512); }
                } else {
                    if (direct) { if (epi) UP_LAUNCH(true, true, 1024); else UP_LAUNCH(true, false, 1024); }
                    else { if (epi) UP_LAUNCH(false, true, 1024); else UP_LAUNCH(false, false, 1024); }
                }
#undef UP_LAUNCH
                HIPCHK(hipGetLastError());
            } else if (halo_can) {
                // rows beyond LDS: a workgroup per row segment, out-of-segment sources prefetched into an LDS halo (k_uppass_halo)
                DevFactor &f0 = *s->upf;
                if (direct && !(s->direct_cache && s->seg_remapped)) {       // the on-the-fly tables were (re)built with columns: positions again
                    s->seg_remapped = true;
                    hipLaunchKernelGGL(k_seg_remap, dim3((unsigned)cdiv(H.DimUp, 256)), dim3(256), 0, s->stream, f0.pelem8.p, f0.pseg8.p, f0.seg_ext.p, f0.seg_ext_off.p, (int32_t)H.DimUp, f0.seglen, f0.seglen, f0.map.p, (int32_t)A.impmask);
                }
                UpHaloArgs U{};
                U.DimUp = A.DimUp; U.DimPh = A.DimPh; U.nseg = f0.nseg; U.seglen = f0.seglen; U.halocap = f0.seg_maxhalo;
                U.Qdw = A.Qdw; U.dw0 = A.dw0; U.nrows = H.Qdw * H.DimPh;
                U.pseg8 = reinterpret_cast<const uint4 *>(f0.pseg8.p); U.ext = f0.seg_ext.p; U.ext_off = f0.seg_ext_off.p; U.tval = f0.ptval.p; U.nterms = f0.pterms;
                U.diag = A.diag; U.Eup = A.Eup; U.nI = A.nI;
                U.v = v; U.out = out; U.ld = (int32_t)ld;
                if (direct) { U.crowtab = s->crow_tab.p; if (!U.crowtab) return fail(EDGPU_EINVAL, "edgpu: internal: row-constant table of the on-the-fly form missing"); }      // built with the other on-the-fly tables (direct_tables)
                const size_t lds = ((size_t)256 + U.seglen + U.halocap + 32) * 8;
                const unsigned grid = (unsigned)std::max(8 * U.nseg, s->up_grid & ~7);
                halo_epi = epi;     // the scale and the -beta*u_prev term move to the dw pass (DwArgs.uprev2): this pass writes the unscaled local part
#define HALO_LAUNCH(D_, Q_, H_, N_)                                                                                             \
    do {                                                                                                                            \
        static OncePerDevice once_;                                                                                                 \
        if (once_.first(s->device)) (void)hipFuncSetAttribute((const void *)k_uppass_halo<D_, Q_, H_, N_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((k_uppass_halo<D_, Q_, H_, N_>), dim3(grid), dim3(1024), lds, s->stream, U);                             \
    } while (0)
#define HALO_N(D_, Q_, H_) do { if (f0.width <= 9) HALO_LAUNCH(D_, Q_, H_, 9); else HALO_LAUNCH(D_, Q_, H_, 10); } while (0)
#define HALO_Q(D_, H_) do { if (nq == 1) HALO_N(D_, 1, H_); else if (nq == 2) HALO_N(D_, 2, H_); else HALO_N(D_, 3, H_); } while (0)
                const int nq = (int)cdiv(U.seglen / 2, 1024);
                if (U.halocap <= 8 * 1024) { if (direct) HALO_Q(true, 8); else HALO_Q(false, 8); }
                else { if (direct) HALO_Q(true, 12); else HALO_Q(false, 12); }
#undef HALO_Q
#undef HALO_N
#undef HALO_LAUNCH
                HIPCHK(hipGetLastError());
            } else if (ld) {
                return fail(EDGPU_EINVAL, "edgpu: internal: pitched vectors on a product form without pitch support");
            } else if (!s->use_lds && plain && s->upf->packed_elem8 && s->up_big) {
                UpBigArgs U{};
                U.DimUp = A.DimUp; U.DimPh = A.DimPh; U.rsplit = s->row_split; U.Qdw = A.Qdw; U.dw0 = A.dw0; U.nrows = H.Qdw * H.DimPh;
                U.packed8 = reinterpret_cast<const uint4 *>(s->upf->pelem8.p); U.tval = s->upf->ptval.p; U.nterms = s->upf->pterms;
                U.diag = A.diag; U.Eup = A.Eup; U.Edw = A.Edw; U.cross = A.cross; U.mapup = A.mapup; U.mapdw = A.mapdw; U.impmask = A.impmask; U.nI = A.nI; U.econst = A.econst;
                U.v = v; U.out = out; U.uprev = uprev; U.scal = s->scal.p;
                const unsigned grid = (unsigned)(cdiv(U.nrows, 8) * 8 * s->row_split);
                if (direct) { if (epi) hipLaunchKernelGGL((k_uppass_big<true, true>), dim3(grid), dim3(512), 0, s->stream, U); else hipLaunchKernelGGL((k_uppass_big<true, false>), dim3(grid), dim3(512), 0, s->stream, U); }
                else { if (epi) hipLaunchKernelGGL((k_uppass_big<false, true>), dim3(grid), dim3(512), 0, s->stream, U); else hipLaunchKernelGGL((k_uppass_big<false, false>), dim3(grid), dim3(512), 0, s->stream, U); }
                HIPCHK(hipGetLastError());
            } else if (!s->use_lds) {
                Kron2Args A2 = A; A2.rsplit = s->row_split;
                CHK(dispatch_kron2(s, base | KF_NODW, A2, s->row_split > 1 ? cdiv(nblocks, 8) * 8 * s->row_split : nblocks));
            } else
                CHK(dispatch_kron2(s, base | KF_NODW | KF_LDS, A, nblocks));
            if (multi) HIPCHK(hipStreamWaitEvent(s->stream, s->ev_gather, 0));
            if (s->tr_on) {      // the dw part came back transposed in tA (DimUp x Qdw): add it, scale, alpha partials
                nblocks = cdiv(H.DimUp, 32) * cdiv(H.Qdw, 32);
                if (epi) hipLaunchKernelGGL(k_tadd<true>, dim3((unsigned)nblocks), dim3(256), 0, s->stream, (const double *)s->tA.p, out, v, H.Qdw, H.DimUp, (const double *)s->scal.p, s->partial.p);
                else hipLaunchKernelGGL(k_tadd<false>, dim3((unsigned)nblocks), dim3(256), 0, s->stream, (const double *)s->tA.p, out, v, H.Qdw, H.DimUp, (const double *)s->scal.p, s->partial.p);
                HIPCHK(hipGetLastError());
            } else if (s->extra && !nd_pass) CHK(dispatch_kron2(s, base | KF_NOLOCAL, A, nblocks));   // phonon cases: row-wise dw pass (+Jx/Jp)
            else {
                if (nd_pass) {
                    NdArgs Nd{};
                    Nd.DimUp = A.DimUp; Nd.DimDw = A.DimDw; Nd.DimPh = A.DimPh; Nd.Norb = A.Norb; Nd.Qdw = A.Qdw; Nd.dw0 = A.dw0;
                    Nd.mapup = A.mapup; Nd.mapdw = A.mapdw; Nd.rankup = A.rankup; Nd.rankdw = A.rankdw; Nd.Jx = A.Jx; Nd.Jp = A.Jp;
                    Nd.vfull = A.vfull; Nd.out = out; Nd.scal = s->scal.p; Nd.xmap = s->halo ? s->xmap_dev.p : nullptr; Nd.ld = (int32_t)ld;
                    if (epi) hipLaunchKernelGGL(k_ndpass<true>, dim3((unsigned)(H.Qdw * H.DimPh)), dim3(256), 0, s->stream, Nd);
                    else hipLaunchKernelGGL(k_ndpass<false>, dim3((unsigned)(H.Qdw * H.DimPh)), dim3(256), 0, s->stream, Nd);
                    HIPCHK(hipGetLastError());
                }
                CHK(run_dw(epi, halo_epi ? uprev : nullptr));
            }
        }
    } else {
        GenArgs A{};
        A.nfac = 2 * H.Ns_Ud; A.DimPh = H.DimPh; A.Norb = d.Norb; A.direct = direct;
        int64_t st = 1;
        for (int k = 0; k < A.nfac; k++) {
            A.dims[k] = H.fac[k].dim; A.strides[k] = st; st *= H.fac[k].dim;
            A.rowptr[k] = s->fac[k]->rowptr.p; A.cols[k] = s->fac[k]->cols.p; A.vals[k] = s->fac[k]->vals.p; A.E[k] = s->fac[k]->E.p; A.maps[k] = s->fac[k]->map.p;
        }
        A.diag = s->diag.p; A.cross = s->cross.p; A.gup = s->gup.p; A.gdw = s->gdw.p; A.nI = 1 << d.Norb; A.total_ud = d.ed_total_ud; A.impmask = (1 << d.Norb) - 1;
        A.econst = H.econst; A.w0 = d.w0_ph; A.gsum = H.gph_sum; A.Nel = s->nloc_el;
        A.NelG = multi ? H.DimUp * H.DimDw : s->nloc_el; A.row0 = multi ? H.dw0 * H.DimUp : 0;
        if (multi) HIPCHK(hipStreamWaitEvent(s->stream, s->ev_gather, 0));   // digits are global: every hop reads the gathered vector
        A.x = multi ? s->vfull.p : v;
        A.v = v; A.out = out; A.uprev = uprev; A.scal = s->scal.p; A.partial = s->partial.p; A.epi = epi;
        nblocks = std::min<int64_t>(cdiv(s->nloc, 256), 4096);
        hipLaunchKernelGGL(k_generic, dim3((unsigned)nblocks), dim3(256), 0, s->stream, A);
        HIPCHK(hipGetLastError());
    }
    if (epi && s->sr_now) { s->sr_na = (int)nblocks; return EDGPU_OK; }      // self-reducing iteration: the axpy kernel sums these partials itself
    if (epi) {
        CHK(reduce_partials(s, nblocks));
        if (!s->defer_ar) CHK(allreduce_tmp(s));      // deferred-norm loop: the caller sums this together with the pending norm
    }
    return EDGPU_OK;
}

}  // namespace edgpu
