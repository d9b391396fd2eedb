// sector_dev.hpp -- device-side construction of a sector: factor images, halo lists, diagonal, explicit CSR; create/destroy
// Part of libedgpu.so (one translation unit: edgpu.hip includes these headers in order).
#pragma once

namespace edgpu {

// ---------------------------------------------------------------------------------------
// device-side construction
// ---------------------------------------------------------------------------------------
// k: index into H.fac (per-orbital / per-spin block), or -1 / -2 for the composite up / dw block of an ed_total_ud=F sector
static int upload_factor(S *s, int k)
{
    const bool comp = k < 0;
    const Factor &f = comp ? s->H.cfac[-1 - k] : s->H.fac[k];
    const bool is_up = comp ? k == -1 : k < s->H.Ns_Ud;
    auto df = std::make_unique<DevFactor>();
    // storage=DIRECT holds no operator image: every hop image and energy table below is derived on the device from the bit patterns
    // by the first product (direct_tables / k_direct_tables*).  The buffers are sized here as for the stored form but their first
    // `dim` entries are filled with a marker (column 0, NaN amplitudes / energies) instead of the host's values, so that an image
    // the table kernels fail to emit shows up as NaN or as a wrong product in every DIRECT parity test.  ed_total_ud=F sectors
    // keep their (composite and per-orbital) tables; the structure-only lists (segment halo lists, x-space row maps) stay.
    const bool derive = !comp && s->H.desc.ed_total_ud != 0 && s->H.desc.storage == EDGPU_STORE_DIRECT;
    const double qnan = std::numeric_limits<double>::quiet_NaN();
    CHK(df->map.upload(f.map)); CHK(df->rank.upload(f.rank));
    std::vector<int32_t> rp32(f.rowptr.begin(), f.rowptr.end());
    CHK(df->rowptr.upload(rp32)); CHK(df->cols.upload(f.cols)); CHK(df->vals.upload(f.vals));
    { std::vector<double> e1(f.E); if (derive) std::fill(e1.begin(), e1.end(), qnan); e1.push_back(0.0); CHK(df->E.upload(e1)); }      // one zero behind the end: the last pair of an odd row (pair kernels on pitched vectors)
    df->width = std::max(f.width, 1);
    for (int32_t i = 0; i < f.dim; i++) if (f.rowptr[i + 1] - f.rowptr[i] < df->width) { df->ragged = true; break; }
    if (df->width > 64) return fail(EDGPU_EUNSUPPORTED, "edgpu: more than 64 hops per spin configuration");
    if ((int)f.terms.size() > kMaxTerms) return fail(EDGPU_EUNSUPPORTED, "edgpu: too many one-body hop terms");
    // ELL image: up factors transposed [width][dim] (lanes run over rows), dw factors [dim][width]
    const bool transposed = is_up || s->tr_on;     // transpose mode: H_dw is applied along rows too
    std::vector<int32_t> ec((size_t)df->width * f.dim); std::vector<double> ev((size_t)df->width * f.dim);
    for (int32_t i = 0; i < f.dim; i++)
        for (int32_t q = 0; q < df->width; q++) {
            int64_t p = f.rowptr[i] + q; bool has = p < f.rowptr[i + 1];
            size_t o = transposed ? (size_t)q * f.dim + i : (size_t)i * df->width + q;
            ec[o] = has ? f.cols[p] : i; ev[o] = has ? f.vals[p] : 0.0;
        }
    if (derive) { std::fill(ec.begin(), ec.end(), 0); std::fill(ev.begin(), ev.end(), qnan); }
    CHK(df->ellc.upload(ec)); CHK(df->ellv.upload(ev));
    // packed image for the persistent local pass: 16-bit column + (sign | term index) byte per hop;
    // padded slots use code 127 -> amplitude 0.  Only when every entry is a single hop term.
    if ((is_up || s->tr_on) && f.dim <= 65536 && f.terms.size() < 127) {      // transpose mode: H_dw is applied along rows too
        const int W = df->width, ncw = (W + 1) / 2, ndw = (W + 3) / 4;
        std::vector<uint32_t> pc((size_t)ncw * f.dim, 0u), pd((size_t)ndw * f.dim, 0u);
        std::vector<double> tvl(128, 0.0);
        for (size_t t = 0; t < f.terms.size(); t++) tvl[t] = std::fabs(f.terms[t].t);
        bool ok = true;
        for (int32_t i = 0; i < f.dim && ok; i++)
            for (int32_t q = 0; q < 4 * ndw; q++) {
                int64_t p = f.rowptr[i] + q; bool has = q < W && p < f.rowptr[i + 1];
                uint32_t col = has ? (uint32_t)f.cols[p] : (uint32_t)i, code = 127u;
                if (has) {
                    int tid = -1;
                    for (size_t t = 0; t < f.terms.size(); t++) if (std::fabs(f.terms[t].t) == std::fabs(f.vals[p])) { tid = (int)t; break; }
                    if (tid < 0) { ok = false; break; }     // summed duplicates: keep the wide image
                    code = (uint32_t)tid | (f.vals[p] < 0.0 ? 0x80u : 0u);
                }
                if (q < 2 * ncw) pc[(size_t)(q / 2) * f.dim + i] |= col << (16 * (q & 1));
                pd[(size_t)(q / 4) * f.dim + i] |= code << (8 * (q & 3));
            }
        // (the element-major images below are composed from pc / pd BEFORE those are cleared for the on-the-fly form)
        auto clear_rows = [&](std::vector<uint32_t> &img, int dwords) { if (derive) std::fill(img.begin(), img.begin() + (size_t)dwords * f.dim, 0u); };
        // odd dim: one more element in the element-major images, a padding configuration without hops (columns = its own index,
        // codes 127): the pair kernels then run odd rows on pitched vectors, whose padding column it stands for
        const int32_t dimp = f.dim + (f.dim & 1);
        if (ok && W <= 10) {                        // element-major image of 8 dwords for rows that do not fit LDS
            std::vector<uint32_t> pe((size_t)8 * dimp, 0u);
            if (dimp > f.dim) { for (int c = 0; c < 5; c++) pe[(size_t)8 * f.dim + c] = (uint32_t)f.dim | ((uint32_t)f.dim << 16); for (int c = 5; c < 8; c++) pe[(size_t)8 * f.dim + c] = 0x7f7f7f7fu; }
            for (int32_t i = 0; i < f.dim; i++) {
                for (int q = 0; q < 10; q++) {
                    int64_t p = f.rowptr[i] + q; bool has = q < W && p < f.rowptr[i + 1];
                    uint32_t col = has ? (uint32_t)f.cols[p] : (uint32_t)i;
                    pe[(size_t)8 * i + q / 2] |= col << (16 * (q & 1));
                }
                for (int q = 0; q < 12; q++) {
                    int64_t p = f.rowptr[i] + q; bool has = q < W && p < f.rowptr[i + 1];
                    uint32_t code = 127u;
                    if (has) { code = (pd[(size_t)(q / 4) * f.dim + i] >> (8 * (q & 3))) & 0xffu; }
                    pe[(size_t)8 * i + 5 + q / 4] |= code << (8 * (q & 3));
                }
            }
            clear_rows(pe, 8); CHK(df->pelem8.upload(pe)); df->packed_elem8 = true;
        }
        if (ok && W <= 16) {                        // element-major image of 12 dwords (8 of columns, 4 of codes: 16 hop slots) for the one-configuration-per-lane kernels:
            std::vector<uint32_t> pe((size_t)12 * f.dim, 0u);      // three 16-byte loads per configuration instead of up to eleven 4-byte ones from the slot-major image
            for (int32_t i = 0; i < f.dim; i++) {
                for (int c = 0; c < 8; c++) pe[(size_t)12 * i + c] = c < ncw ? pc[(size_t)c * f.dim + i] : ((uint32_t)i | ((uint32_t)i << 16));
                for (int c = 0; c < 4; c++) pe[(size_t)12 * i + 8 + c] = c < ndw ? pd[(size_t)c * f.dim + i] : 0x7f7f7f7fu;
            }
            clear_rows(pe, 12); CHK(df->pelem16.upload(pe)); df->packed_elem16 = true;
        }
        if (ok && W <= 8) {                         // element-major image for the pair kernel: 6 dwords per up configuration
            std::vector<uint32_t> pe((size_t)6 * dimp, 0u);
            if (dimp > f.dim) { for (int c = 0; c < 4; c++) pe[(size_t)6 * f.dim + c] = (uint32_t)f.dim | ((uint32_t)f.dim << 16); for (int c = 4; c < 6; c++) pe[(size_t)6 * f.dim + c] = 0x7f7f7f7fu; }
            for (int32_t i = 0; i < f.dim; i++) {
                for (int c = 0; c < 4; c++) pe[(size_t)6 * i + c] = c < ncw ? pc[(size_t)c * f.dim + i] : ((uint32_t)i | ((uint32_t)i << 16));
                for (int c = 0; c < 2; c++) pe[(size_t)6 * i + 4 + c] = c < ndw ? pd[(size_t)c * f.dim + i] : 0x7f7f7f7fu;
            }
            clear_rows(pe, 6); CHK(df->pelem.upload(pe)); df->packed_elem = true;
        }
        if (ok) { if (derive) { std::fill(pc.begin(), pc.end(), 0u); std::fill(pd.begin(), pd.end(), 0u); } CHK(df->pcols.upload(pc)); CHK(df->pcodes.upload(pd)); CHK(df->ptval.upload(tvl)); df->packed = true; df->pterms = 128; }
    }
    std::vector<int32_t> ta, tb; std::vector<double> tv;
    for (auto &t : f.terms) { ta.push_back(t.alpha); tb.push_back(t.beta); tv.push_back(t.t); }
    if (ta.empty()) { ta.push_back(1); tb.push_back(1); tv.push_back(0.0); }
    CHK(df->talpha.upload(ta)); CHK(df->tbeta.upload(tb)); CHK(df->tval.upload(tv));
    // per-level one-body coefficients for the on-the-fly energy table
    if (comp) { std::vector<double> lin(std::max(f.nbits, 1), 0.0); CHK(df->lin.upload(lin)); }     // composite tables are not rebuilt on the fly
    else {
        const edgpu_desc &d = s->H.desc; const SectorHost &H = s->H;
        int spin = k < H.Ns_Ud ? 1 : d.Nspin;
        std::vector<double> lin(f.nbits, 0.0);
        auto hf = [&](int a) { return d.hfmode ? -0.5 * d.Uloc[a - 1] - (d.Norb - 1) * (0.5 * d.Ust + 0.5 * (d.Ust - d.Jh)) : 0.0; };
        if (d.ed_total_ud) {
            for (int a = 1; a <= d.Norb; a++) lin[a - 1] += H.imp(spin, spin, a, a) - d.xmu + hf(a);
            for (int a = 1; a <= d.Nfoo; a++) for (int kp = 1; kp <= d.Nbath; kp++) lin[H.stride[a - 1][kp - 1] - 1] += H.bdiag(spin, a, kp);
        } else {
            int a = (k % H.Ns_Ud) + 1;
            lin[0] += H.imp(spin, spin, a, a) - d.xmu + hf(a);
            for (int kp = 1; kp <= d.Nbath; kp++) lin[kp] += H.bdiag(spin, a, kp);
        }
        CHK(df->lin.upload(lin));
    }
    if (comp) s->cdev[-1 - k] = std::move(df); else s->fac.push_back(std::move(df));
    return EDGPU_OK;
}

// Segments of the up factor for rows beyond LDS (k_uppass_halo): smallest power-of-two segment count whose segment and
// halo list fit the kernel's register staging and 160 KB of LDS.  Structure only (which columns the hops reach), so the
// lists serve the stored and the on-the-fly product alike.
static int build_up_segments(S *s)
{
    const Factor &f = *s->hup; DevFactor &df = *s->upf;
    if (!df.packed_elem8 || f.dim % 2 || f.dim > 65536) return EDGPU_OK;
    // first choice: two pairs and eight halo values per lane (the kernel variant that fits 128 registers without spilling)
    const int32_t lim[2][2] = {{std::min(knob("EDGPU_UP_SEGMAX"), kHaloSegMax), std::min(XKNOB("EDGPU_UP_HALOMAX", 8192), kHaloMax)}, {kHaloSegMax, kHaloMax}};
    for (int pass = 0; pass < 2; pass++)
        for (int nseg = 2; nseg <= 64; nseg *= 2) {
            const int32_t seglen = (int32_t)((cdiv(f.dim, nseg) + 1) & ~(int64_t)1);
            if (seglen > lim[pass][0]) continue;
            std::vector<int32_t> ext, off(1, 0); int32_t maxh = 0; bool ok = true;
            for (int sg = 0; sg < nseg && ok; sg++) {
                const int32_t lo = std::min<int32_t>(sg * seglen, f.dim), hi = std::min<int32_t>(lo + seglen, f.dim);
                std::vector<int32_t> e;
                for (int32_t i = lo; i < hi; i++)
                    for (int64_t p = f.rowptr[i]; p < f.rowptr[i + 1]; p++) if (f.cols[p] < lo || f.cols[p] >= hi) e.push_back(f.cols[p]);
                std::sort(e.begin(), e.end()); e.erase(std::unique(e.begin(), e.end()), e.end());
                if ((int64_t)e.size() > lim[pass][1]) ok = false;
                maxh = std::max<int32_t>(maxh, (int32_t)e.size());
                ext.insert(ext.end(), e.begin(), e.end()); off.push_back((int32_t)ext.size());
            }
            if (!ok || ((size_t)seglen + maxh + 256 + 32 + 16) * 8 > 156 * 1024) continue;      // + signed amplitude table, row constants
            if (ext.empty()) ext.push_back(0);
            CHK(df.seg_ext.upload(ext)); CHK(df.seg_ext_off.upload(off)); CHK(df.pseg8.alloc((size_t)8 * f.dim));
            df.nseg = nseg; df.seglen = seglen; df.seg_maxhalo = maxh; df.seg_ok = true;
            hipLaunchKernelGGL(k_seg_remap, dim3((unsigned)cdiv(f.dim, 256)), dim3(256), 0, s->stream, df.pelem8.p, df.pseg8.p, df.seg_ext.p, df.seg_ext_off.p, f.dim, seglen, seglen, df.map.p, s->H.desc.ed_total_ud ? (1 << s->H.desc.Norb) - 1 : 0);
            HIPCHK(hipGetLastError());
            return EDGPU_OK;
        }
    return EDGPU_OK;
}

// Which dw rows does each rank need from each other rank?  Row j of rank r is needed here iff some
// local row i has H_dw[i,j] != 0 -- or, with Jx/Jp, j is the partner row of i (one impurity electron of the dw
// word moved to another orbital, stored/H_non_local.f90:21-83).  H_dw is small and known to every rank, so the lists are
// computed locally on both sides (no set-up communication).  At Ns=16 this is 53 % / 36 % / 24 %
// of the all-gather volume at 2 / 4 / 8 ranks.
// Relay routing of the needed-rows exchange (opt-in).  The partner blocks of a dw block are its hypercube neighbours, so a few
// of a rank's links carry most of what it receives while the others idle (Ns=18 on 8 ranks: 3 548 of 10 181 rows from one peer).
// Two phases: in phase 1 every pair sends half of its direct rows, and the heavy pairs hand a chunk of theirs to a third rank;
// in phase 2 the other half travels and the relays forward what they hold.  At most one relayed chunk per directed link and
// phase; greedy: take the pair with the largest load, give it the relay that lowers it most, until nothing improves.  Every rank
// runs the same integer plan on the same matrix.  Row order of pair (s -> d) in s's send buffer and in d's x-space:
// [phase-1 direct | phase-2 direct | chunk 1 | chunk 2 ...].  Model: scripts/relay_model.py (busiest link 3 548 -> 2 622 rows
// at Ns=18 x 8, 954 -> 662 at Ns=16 x 8).  Unmeasured on xGMI, like the rest of the multi-rank code.
static int relay_plan(S *s)
{
    const int P = s->comm.size, me = s->comm.rank;
    const std::vector<int64_t> &M = s->xch_rows;
    auto at = [P](std::vector<int64_t> &v, int a, int b) -> int64_t & { return v[(size_t)a * P + b]; };
    std::vector<int64_t> dirn = M, rel1((size_t)P * P, 0), rel2((size_t)P * P, 0);
    std::vector<char> used1((size_t)P * P, 0), used2((size_t)P * P, 0);
    struct Chunk { int s, d, r; int64_t cnt; };
    std::vector<Chunk> chunks;
    auto d1 = [&](int a, int b) { return (dirn[(size_t)a * P + b] + 1) / 2; };
    auto d2 = [&](int a, int b) { return dirn[(size_t)a * P + b] - d1(a, b); };
    auto L1 = [&](int a, int b) { return d1(a, b) + rel1[(size_t)a * P + b]; };
    auto L2 = [&](int a, int b) { return d2(a, b) + rel2[(size_t)a * P + b]; };
    auto total = [&]() { int64_t m1 = 0, m2 = 0; for (int a = 0; a < P; a++) for (int b = 0; b < P; b++) { m1 = std::max(m1, L1(a, b)); m2 = std::max(m2, L2(a, b)); } return m1 + m2; };
    s->relay_before = 0; for (int64_t v : M) s->relay_before = std::max(s->relay_before, v);
    for (int it = 0; it < 4 * P * P; it++) {
        int bs = -1, bd = -1; int64_t Ld = 0;
        for (int a = 0; a < P; a++) for (int b = 0; b < P; b++) { const int64_t l = std::max(L1(a, b), L2(a, b)); if (at(dirn, a, b) > 0 && l > Ld) { Ld = l; bs = a; bd = b; } }
        if (bs < 0) break;
        int br = -1; int64_t bdelta = 0;
        for (int r = 0; r < P; r++) {
            if (r == bs || r == bd || used1[(size_t)bs * P + r] || used2[(size_t)r * P + bd]) continue;
            const int64_t base = std::max(L1(bs, r), L2(r, bd));
            int64_t delta = (Ld - base) * 2 / 3;
            delta = std::min(delta, at(dirn, bs, bd));
            if (delta >= 1 && delta > bdelta) { bdelta = delta; br = r; }
        }
        if (br < 0) break;
        const int64_t before = total();
        at(dirn, bs, bd) -= bdelta; at(rel1, bs, br) += bdelta; at(rel2, br, bd) += bdelta;
        if (total() > before) { at(dirn, bs, bd) += bdelta; at(rel1, bs, br) -= bdelta; at(rel2, br, bd) -= bdelta; break; }      // no longer an improvement
        used1[(size_t)bs * P + br] = 1; used2[(size_t)br * P + bd] = 1;
        chunks.push_back({bs, bd, br, bdelta});
    }
    s->relay_after = total(); s->relay_chunks = (int)chunks.size();
    s->relay_link_rows.assign((size_t)P * P, 0);
    for (int a = 0; a < P; a++) for (int b = 0; b < P; b++) s->relay_link_rows[(size_t)a * P + b] = L1(a, b) + L2(a, b);
    // this rank's pieces
    s->relay.assign((size_t)P, S::RelayPeer{});
    for (int b = 0; b < P; b++) {
        if (b == me) continue;
        S::RelayPeer &Q = s->relay[b];
        Q.s1d = {s->send_rows_off[b], d1(me, b)}; Q.s2d = {s->send_rows_off[b] + d1(me, b), d2(me, b)};
        Q.r1d = {s->recv_rows_off[b], d1(b, me)}; Q.r2d = {s->recv_rows_off[b] + d1(b, me), d2(b, me)};
    }
    std::vector<int64_t> coff((size_t)P * P, 0);      // rows of pair (s,d) placed so far behind its direct rows
    int64_t stage = 0;
    for (const Chunk &c : chunks) {
        const int64_t off = at(dirn, c.s, c.d) + at(coff, c.s, c.d);
        at(coff, c.s, c.d) += c.cnt;
        if (c.s == me) s->relay[c.r].s1r = {s->send_rows_off[c.d] + off, c.cnt};               // hand to the relay, from my send block for d
        if (c.r == me) { s->relay[c.s].r1r = {stage, c.cnt}; s->relay[c.d].s2f = {stage, c.cnt}; stage += c.cnt; }      // hold, then forward
        if (c.d == me) s->relay[c.r].r2f = {s->recv_rows_off[c.s] + off, c.cnt};               // lands behind the direct rows of s
    }
    CHK(s->relay_stage.alloc((size_t)std::max<int64_t>(stage, 1) * (size_t)((s->H.DimUp + 15) & ~(int64_t)15)));
    s->relay_on = true;
    if ((s->dbg & 8) && me == 0) {
        fprintf(stderr, "edgpu: relay routing on %d ranks: busiest link %lld rows -> %lld (two phases), %d relayed chunks\n", P, (long long)s->relay_before, (long long)s->relay_after, s->relay_chunks);
        for (int a = 0; a < P; a++) { fprintf(stderr, "  rows %d ->:", a); for (int b = 0; b < P; b++) fprintf(stderr, " %lld/%lld", (long long)M[(size_t)a * P + b], (long long)s->relay_link_rows[(size_t)a * P + b]); fprintf(stderr, "\n"); }
    }
    return EDGPU_OK;
}

static int setup_halo(S *s)
{
    const SectorHost &H = s->H; const Factor &dw = *s->hdw;
    const int P = s->comm.size, me = s->comm.rank;
    auto range_of = [&](int r, int64_t &d0, int64_t &qd) { int64_t q = H.DimDw / P, rem = H.DimDw % P; qd = r < rem ? q + 1 : q; d0 = r < rem ? r * (q + 1) : r * q + rem; };
    std::vector<int> owner(H.DimDw);
    for (int r = 0; r < P; r++) { int64_t d0, qd; range_of(r, d0, qd); for (int64_t j = d0; j < d0 + qd; j++) owner[j] = r; }
    std::vector<std::vector<int32_t>> recv_rows(P), send_rows(P);
    {
        std::vector<char> mark(H.DimDw, 0);
        const edgpu_desc &d = H.desc;
        auto mark_row = [&](int64_t i) {       // rows that row i reads
            for (int64_t p = dw.rowptr[i]; p < dw.rowptr[i + 1]; p++) mark[dw.cols[p]] = 1;
            if (!H.Jhflag) return;
            const uint32_t m = (uint32_t)dw.map[i];
            for (int a = 0; a < d.Norb; a++)
                for (int b = 0; b < d.Norb; b++) {
                    if (a == b) continue;
                    const uint32_t ba = 1u << a, bb = 1u << b;
                    if ((m & ba) && !(m & bb)) { const int32_t j = dw.rank[(m & ~ba) | bb]; if (j >= 0) mark[j] = 1; }
                }
        };
        for (int64_t i = H.dw0; i < H.dw0 + H.Qdw; i++) mark_row(i);
        for (int64_t j = 0; j < H.DimDw; j++) if (mark[j] && owner[j] != me) recv_rows[owner[j]].push_back((int32_t)j);
        // what rank r needs from me: rows j of mine that some row i of r couples to; and, for the routing of the exchange, how
        // many rows every rank needs from every other one (xch_rows[s*P+d], the same on all ranks)
        s->xch_rows.assign((size_t)P * P, 0);
        for (int r = 0; r < P; r++) {
            int64_t d0, qd; range_of(r, d0, qd);
            std::fill(mark.begin(), mark.end(), 0);
            for (int64_t i = d0; i < d0 + qd; i++) mark_row(i);
            for (int64_t j = 0; j < H.DimDw; j++) if (mark[j] && owner[j] != r) s->xch_rows[(size_t)owner[j] * P + r]++;
            if (r == me) continue;
            for (int64_t j = H.dw0; j < H.dw0 + H.Qdw; j++) if (mark[j]) send_rows[r].push_back((int32_t)(j - H.dw0));
        }
    }
    s->send_rows_cnt.assign(P, 0); s->recv_rows_cnt.assign(P, 0); s->send_rows_off.assign(P, 0); s->recv_rows_off.assign(P, 0);
    std::vector<int32_t> send_list, xmap(H.DimDw, -1);
    int64_t so = 0, ro = 0;
    for (int64_t q = 0; q < H.Qdw; q++) xmap[H.dw0 + q] = (int32_t)q;
    for (int r = 0; r < P; r++) {
        s->send_rows_cnt[r] = (int64_t)send_rows[r].size(); s->send_rows_off[r] = so; so += s->send_rows_cnt[r];
        s->recv_rows_cnt[r] = (int64_t)recv_rows[r].size(); s->recv_rows_off[r] = ro;
        for (size_t k = 0; k < recv_rows[r].size(); k++) xmap[recv_rows[r][k]] = (int32_t)(H.Qdw + ro + (int64_t)k);
        ro += s->recv_rows_cnt[r];
        send_list.insert(send_list.end(), send_rows[r].begin(), send_rows[r].end());
    }
    s->nhalo = H.Qdw + ro;
    if (send_list.empty()) send_list.push_back(0);
    CHK(s->send_list.upload(send_list));
    {   // message rows may carry the line-aligned pitch of the explicit CSR (decided later, in build_csr): room for it, zeros in the padding
        const size_t nsb = (size_t)std::max<int64_t>(so, 1) * (size_t)((H.DimUp + 15) & ~(int64_t)15) * H.DimPh;
        CHK(s->sendbuf.alloc(nsb)); HIPCHK(hipMemsetAsync(s->sendbuf.p, 0, nsb * 8, s->stream));
    }
    // H_dw column indices translated to x-space for the local rows (ELL image and CSR image)
    const int W = s->dwf->width;
    std::vector<int32_t> ec((size_t)W * dw.dim, 0), cc(dw.cols.size(), 0);
    for (int64_t i = H.dw0; i < H.dw0 + H.Qdw; i++) {
        for (int q = 0; q < W; q++) { int64_t p = dw.rowptr[i] + q; bool has = p < dw.rowptr[i + 1]; ec[(size_t)i * W + q] = has ? xmap[dw.cols[p]] : xmap[i]; }
        for (int64_t p = dw.rowptr[i]; p < dw.rowptr[i + 1]; p++) cc[p] = xmap[dw.cols[p]];
    }
    CHK(s->dw_ellc_x.upload(ec)); CHK(s->dw_cols_x.upload(cc));
    if (H.Jhflag) CHK(s->xmap_dev.upload(xmap));      // the non-local pass finds its partner rows through this
    HIPCHK(hipEventCreateWithFlags(&s->ev_pack, hipEventDisableTiming));
    s->halo = true;
    if (P >= 3 && H.DimPh == 1 && knob("EDGPU_RELAY") != 0) CHK(relay_plan(s));
    return EDGPU_OK;
}

static int build_diag(S *s)
{
    const SectorHost &H = s->H; const edgpu_desc &d = H.desc;
    DiagArgs A{};
    A.Norb = d.Norb; A.Nspin = d.Nspin; A.Nbath = d.Nbath; A.Nfoo = d.Nfoo; A.hfmode = d.hfmode; A.total_ud = d.ed_total_ud; A.Ns_Ud = H.Ns_Ud;
    for (int k = 0; k < 2 * H.Ns_Ud; k++) { A.dims[k] = H.fac[k].dim; A.maps[k] = s->fac[k]->map.p; }
    A.nloc = s->nloc_el; A.shift = H.dw0 * H.DimUp;
    for (int a = 0; a < EDGPU_MAX_ORB; a++) { A.Uloc[a] = d.Uloc[a]; A.imp_up[a] = A.imp_dw[a] = 0.0; }
    for (int a = 1; a <= d.Norb; a++) { A.imp_up[a - 1] = H.imp(1, 1, a, a); A.imp_dw[a - 1] = H.imp(d.Nspin, d.Nspin, a, a); }
    A.Ust = d.Ust; A.Jh = d.Jh; A.xmu = d.xmu;
    A.bd_up = s->bd_up.p; A.bd_dw = s->bd_dw.p; A.stride = s->stride.p;
    CHK(s->diag.alloc((size_t)s->nloc_el + 2)); HIPCHK(hipMemsetAsync(s->diag.p + s->nloc_el, 0, 16, s->stream));      // + 2: the last pair of an odd last row reads one element behind the end
    hipLaunchKernelGGL(k_build_diag, dim3((unsigned)cdiv(s->nloc_el, 256)), dim3(256), 0, s->stream, A, s->diag.p);
    HIPCHK(hipGetLastError());
    return EDGPU_OK;
}

// row pitch of the x-space [own rows | received rows] and of the vectors the exchange reads: padded only for the explicit CSR whose
// columns were built in the pitched space (build_csr)
static inline int64_t xspace_ld(const S *s) { return (s->H.desc.storage == EDGPU_STORE_CSR && s->halo && s->csr_ld > s->H.DimUp) ? s->csr_ld : s->H.DimUp; }
static bool fused_ok(const S *s);      // product.hpp
static bool pitch_ok(const S *s);

// Non-local (Jx/Jp) block of the explicit storage: count sweep, host scan over the dw rows, fill sweep (k_nd_build)
static int build_csr_nd(S *s)
{
    const SectorHost &H = s->H; const edgpu_desc &d = H.desc;
    NdBuildArgs A{};
    A.DimUp = (int32_t)H.DimUp; A.DimDw = (int32_t)(s->halo ? s->nhalo : H.DimDw); A.Norb = d.Norb; A.Qdw = H.Qdw; A.dw0 = H.dw0;
    A.mapup = s->upf->map.p; A.mapdw = s->dwf->map.p; A.rankup = s->upf->rank.p; A.rankdw = s->dwf->rank.p; A.xmap = s->halo ? s->xmap_dev.p : nullptr;
    A.Jx = d.Jx; A.Jp = d.Jp;
    DevBuf<int64_t> rnr, rne;
    CHK(rnr.alloc((size_t)H.Qdw)); CHK(rne.alloc((size_t)H.Qdw));
    A.row_nr = rnr.p; A.row_ne = rne.p;
    hipLaunchKernelGGL(k_nd_build<0>, dim3((unsigned)H.Qdw), dim3(256), 0, s->stream, A);
    HIPCHK(hipGetLastError());
    std::vector<int64_t> nr((size_t)H.Qdw), ne((size_t)H.Qdw);
    HIPCHK(hipMemcpyAsync(nr.data(), rnr.p, (size_t)H.Qdw * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(ne.data(), rne.p, (size_t)H.Qdw * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    int64_t tr = 0, te = 0;
    for (int64_t q = 0; q < H.Qdw; q++) { const int64_t a = nr[q], b = ne[q]; nr[q] = tr; ne[q] = te; tr += a; te += b; }
    s->nd_nr = tr; s->nd_nnz = te;
    CHK(s->nd_rows.alloc((size_t)tr + 1)); CHK(s->nd_ptr.alloc((size_t)tr + 1)); CHK(s->nd_cols.alloc((size_t)te + 1)); CHK(s->nd_vals.alloc((size_t)te + 1)); CHK(s->nd_slot.alloc((size_t)te + 1));
    HIPCHK(hipMemcpyAsync(rnr.p, nr.data(), (size_t)H.Qdw * 8, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(rne.p, ne.data(), (size_t)H.Qdw * 8, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->nd_ptr.p + tr, &te, 8, hipMemcpyHostToDevice, s->stream));
    A.rows = s->nd_rows.p; A.ptr = s->nd_ptr.p; A.cols = s->nd_cols.p; A.vals = s->nd_vals.p; A.slot = s->nd_slot.p;
    hipLaunchKernelGGL(k_nd_build<1>, dim3((unsigned)H.Qdw), dim3(256), 0, s->stream, A);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));      // rnr / rne and the host copy of `te` go out of scope
    return EDGPU_OK;
}

static int build_csr(S *s)
{
    const SectorHost &H = s->H;
    if (!s->kron2) return fail(EDGPU_EUNSUPPORTED, "edgpu: storage=CSR needs the two-factor form of the sector (composite blocks too large)");
    if (H.DimUp * s->nhalo >= ((int64_t)1 << 31)) return fail(EDGPU_EUNSUPPORTED, "edgpu: storage=CSR needs DimUp*(rows of x) < 2^31 (int32 columns)");
    const Factor &up = *s->hup, &dw = *s->hdw;
    std::vector<int64_t> uppre(up.dim + 1, 0), dwpre(H.Qdw + 1, 0);
    for (int i = 0; i < up.dim; i++) uppre[i + 1] = uppre[i] + 1 + (up.rowptr[i + 1] - up.rowptr[i]);
    int64_t maxrow = 0; int upmax = 0;
    for (int i = 0; i < up.dim; i++) upmax = std::max<int>(upmax, (int)(up.rowptr[i + 1] - up.rowptr[i]));
    for (int64_t q = 0; q < H.Qdw; q++) {
        int64_t nd = dw.rowptr[H.dw0 + q + 1] - dw.rowptr[H.dw0 + q];
        dwpre[q + 1] = dwpre[q] + uppre[up.dim] + nd * up.dim;
        maxrow = std::max<int64_t>(maxrow, 1 + upmax + nd);
    }
    s->csr_nnz = dwpre[H.Qdw];
    CHK(s->uppre.upload(uppre)); CHK(s->dwpre.upload(dwpre));
    // Two storage forms of the same explicit matrix: one CSR, or (default for large sectors) two CSR
    // blocks H = H_loc + H_dw so that each block can be streamed in the order that keeps ITS gathers
    // cache resident (rows for the up hops, XCD-owned column tiles for the dw hops).
    { const int k = knob("EDGPU_CSR_SPLIT"); s->csr_split = (k < 0 ? H.DimUp * H.DimDw * 8 > ((int64_t)12 << 20) /* measured: 16 MB 130 -> 109 us per step, 32 MB 253 -> 218, 48 MB 362 -> 310, tie at 5-8 MB, the single block wins below (profiles/r04/kron_split_threshold.txt); 48 MiB until round 4 */ : k != 0) || H.Jhflag || H.DimPh > 1; }      // the non-local / phonon passes go between the local and the dw block
    s->csrL_nnz = H.Qdw * uppre[up.dim]; s->csrD_nnz = s->csr_nnz - s->csrL_nnz;
    s->csr_ld = H.DimUp;
    // several ranks: only with the needed-rows exchange, whose messages carry the pitch (EDGPU_CSR_PITCH_MULTI=0: dense rows as in round 2)
    if ((!s->multi || (s->halo && knob("EDGPU_CSR_PITCH_MULTI") != 0)) && s->csr_split && !H.Jhflag && H.DimPh == 1 && (H.DimUp % 16) != 0 && knob("EDGPU_PAD") != 0 && knob("EDGPU_CSR_PITCH") != 0) {
        const int64_t ld = (H.DimUp + 15) & ~(int64_t)15;
        if (ld * s->nhalo < ((int64_t)1 << 31)) s->csr_ld = ld;
    }
    CsrBuildArgs A{};
    A.ld = s->csr_ld;
    A.DimUp = (int32_t)H.DimUp; A.DimDw = (int32_t)H.DimDw; A.Qdw = H.Qdw; A.dw0 = H.dw0;
    A.uprp = s->upf->rowptr.p; A.upc = s->upf->cols.p; A.upv = s->upf->vals.p;
    A.dwrp = s->dwf->rowptr.p; A.dwc = s->halo ? s->dw_cols_x.p : s->dwf->cols.p; A.dwv = s->dwf->vals.p;
    A.rowshift = s->halo ? H.dw0 : 0;     // x-space index of local row idw is idw - rowshift
    A.uppre = s->uppre.p; A.dwpre = s->dwpre.p; A.diag = s->diag.p;
    const unsigned gb = (unsigned)cdiv(s->nloc_el + 1, 256);
    if (!s->csr_split) {
        CHK(s->csr_rowptr.alloc((size_t)s->nloc_el + 1)); CHK(s->csr_cols.alloc((size_t)s->csr_nnz + 8)); CHK(s->csr_vals.alloc((size_t)s->csr_nnz + 8));
        A.rowptr = s->csr_rowptr.p; A.cols = s->csr_cols.p; A.vals = s->csr_vals.p;
        hipLaunchKernelGGL(k_csr_build, dim3(gb), dim3(256), 0, s->stream, A, 0);
    } else {
        CHK(s->csr_rowptr.alloc((size_t)s->nloc_el + 1)); CHK(s->csr_cols.alloc((size_t)s->csrL_nnz + 8)); CHK(s->csr_vals.alloc((size_t)s->csrL_nnz + 8));
        {   // lane-major block images (k_csr_il): 64-row blocks whose rows all have the same length
            bool up_uniform = true;
            for (int i = 1; i < up.dim; i++) up_uniform = up_uniform && (up.rowptr[i + 1] - up.rowptr[i] == up.rowptr[1] - up.rowptr[0]);
            const int il = knob("EDGPU_CSR_IL");      // bit 0: local block, bit 1: dw-type blocks
            s->csr_il_loc = (il & 1) && (up_uniform || knob("EDGPU_CSR_JAG") != 0);      // rows of different lengths: packed level by level (k_csr_il, jagged path)
            s->csr_il_dw = (il & 2) != 0;
        }
        A.rpb = 64;
        s->csr_three = s->halo && !H.Jhflag && knob("EDGPU_CSR_THREE") != 0;      // the stored non-local block reads own and received rows from one x buffer: keep the own-row copy
        // per local dw row: its dw hops into the rank's own rows (x-space row < Qdw; three-block form only) and the others.
        // Every row of one idw row holds the same number of them: factored row pointers base[idw] + iup * cnt[idw].
        std::vector<int64_t> bD(H.Qdw + 1, 0), bO(H.Qdw + 1, 0), pD(H.Qdw + 1, 0), pO(H.Qdw + 1, 0); std::vector<int32_t> cD(H.Qdw, 0), cO(H.Qdw, 0);
        {
            std::vector<int32_t> xc;
            if (s->csr_three) { xc.resize(dw.cols.size()); HIPCHK(hipMemcpy(xc.data(), s->dw_cols_x.p, xc.size() * 4, hipMemcpyDeviceToHost)); }
            for (int64_t q = 0; q < H.Qdw; q++) {
                int32_t no = 0, nh = 0;
                for (int64_t pp = dw.rowptr[H.dw0 + q]; pp < dw.rowptr[H.dw0 + q + 1]; pp++) (s->csr_three && xc[pp] < H.Qdw ? no : nh)++;
                cD[q] = nh; cO[q] = no; pD[q + 1] = pD[q] + nh; pO[q + 1] = pO[q] + no;
                bD[q + 1] = bD[q] + (int64_t)nh * up.dim; bO[q + 1] = bO[q] + (int64_t)no * up.dim;
            }
        }
        s->csr_uni = knob("EDGPU_CSR_UNI") != 0;
        // tile-major storage of the lane-major dw-type blocks: the blocks of one column tile (all idw) are contiguous, in the
        // order the XCD that owns the tile walks them -- block (jb, idw) starts at jb*64*ctot + rows(jb)*cpre[idw]
        s->csr_tm = s->csr_il_dw && s->csr_uni && knob("EDGPU_CSR_TM") != 0;
        s->h_cntD = cD; s->h_cntO = cO;
        CHK(s->csrD_base.upload(bD)); CHK(s->csrD_cnt.upload(cD)); CHK(s->csrD_cpre.upload(pD));
        if (s->csr_three) { CHK(s->csrO_base.upload(bO)); CHK(s->csrO_cnt.upload(cO)); CHK(s->csrO_cpre.upload(pO)); s->csrO_nnz = bO[H.Qdw]; s->csrD_nnz = bD[H.Qdw]; }
        CHK(s->csrD_rowptr.alloc((size_t)s->nloc_el + 1)); CHK(s->csrD_cols.alloc((size_t)s->csrD_nnz + 8)); CHK(s->csrD_vals.alloc((size_t)s->csrD_nnz + 8));
        if (s->csr_three) { CHK(s->csrO_rowptr.alloc((size_t)s->nloc_el + 1)); CHK(s->csrO_cols.alloc((size_t)s->csrO_nnz + 8)); CHK(s->csrO_vals.alloc((size_t)s->csrO_nnz + 8)); }
        A.rowptr = s->csr_rowptr.p; A.cols = s->csr_cols.p; A.vals = s->csr_vals.p;
        A.il = s->csr_il_loc ? 1 : 0;
        hipLaunchKernelGGL(k_csr_build, dim3(gb), dim3(256), 0, s->stream, A, 1);
        A.il = s->csr_il_dw ? 1 : 0; A.tm = s->csr_tm ? 1 : 0;
        if (!s->csr_three) {
            A.rowptr = s->csrD_rowptr.p; A.cols = s->csrD_cols.p; A.vals = s->csrD_vals.p; A.cpre = s->csrD_cpre.p;
            hipLaunchKernelGGL(k_csr_build, dim3(gb), dim3(256), 0, s->stream, A, 2);
        } else {
            A.sel = 1; A.selpre = s->csrO_base.p; A.cpre = s->csrO_cpre.p; A.rowptr = s->csrO_rowptr.p; A.cols = s->csrO_cols.p; A.vals = s->csrO_vals.p;
            hipLaunchKernelGGL(k_csr_build, dim3(gb), dim3(256), 0, s->stream, A, 2);
            A.sel = 2; A.selpre = s->csrD_base.p; A.cpre = s->csrD_cpre.p; A.rowptr = s->csrD_rowptr.p; A.cols = s->csrD_cols.p; A.vals = s->csrD_vals.p;
            hipLaunchKernelGGL(k_csr_build, dim3(gb), dim3(256), 0, s->stream, A, 2);
            A.sel = 0; A.selpre = nullptr;
        }
        A.il = 0; A.tm = 0;
        int64_t ndmax = 0;
        for (int64_t q = 0; q < H.Qdw; q++) ndmax = std::max<int64_t>(ndmax, dw.rowptr[H.dw0 + q + 1] - dw.rowptr[H.dw0 + q]);
        s->csrD_rpb = s->csr_il_dw ? 64 : (int32_t)std::min<int64_t>(knob("EDGPU_CSRD_ROWS"), 1152 / std::max<int64_t>(ndmax, 1));
        if (s->csrD_rpb < 1) return fail(EDGPU_EUNSUPPORTED, "edgpu: a CSR row exceeds the LDS tile");
        s->csrD_bpr = (int32_t)cdiv(H.DimUp, s->csrD_rpb);
        s->csrD_nt = knob("EDGPU_CSRD_NT"); s->csrL_nt = XKNOB("EDGPU_CSRL_NT", 0) != 0;      // measured: plain loads for the row-ordered local block (8.12 vs 8.43 ms per product at c3), non-temporal ones for the tile-ordered dw block
        maxrow = 1 + upmax;
        s->csrL_small = (1 + upmax) * 64 <= 640 && XKNOB("EDGPU_CSR_SMALL", 1);     // smaller LDS image -> more resident workgroups
        s->csrD_small = ndmax * s->csrD_rpb <= 640 && XKNOB("EDGPU_CSR_SMALL", 1);
    }
    HIPCHK(hipGetLastError());
    s->csr_xcd_rows = XKNOB("EDGPU_CSR_XCD", 0);
    s->csr_rpb = s->csr_il_loc ? 64 : (int32_t)std::min<int64_t>(64, 1152 / std::max<int64_t>(maxrow, 1));      // one wave per row block
    if (s->csr_rpb < 1) return fail(EDGPU_EUNSUPPORTED, "edgpu: a CSR row exceeds the LDS tile");
    s->csr_bpr = (int32_t)cdiv(H.DimUp, s->csr_rpb);
    s->csr_tile = std::max(1, std::min(XKNOB("EDGPU_CSR_TILE", 1 << 20), s->csr_bpr));
    if (H.Jhflag) CHK(build_csr_nd(s));
    if (H.DimPh > 1) {
        CHK(s->csr_gel.alloc((size_t)s->nloc_el));
        hipLaunchKernelGGL(k_build_gel, dim3((unsigned)std::min<int64_t>(cdiv(s->nloc_el, 256), 65536)), dim3(256), 0, s->stream, s->nloc_el, (int32_t)H.DimUp, H.dw0,
                           s->upf->map.p, s->dwf->map.p, (int32_t)((1 << H.desc.Norb) - 1), s->gup.p, s->gdw.p, H.gph_sum, s->csr_gel.p);
        HIPCHK(hipGetLastError());
    }
    return EDGPU_OK;
}

static int64_t max_partials(const S *s)
{
    int64_t n = std::max<int64_t>(s->H.Qdw * s->H.DimPh, 4096);
    if (s->kron2) {   // column-tiled dw pass: one partial per workgroup
        const int w = s->dw_w, R = std::min(256 / w, s->dw_rows);
        n = std::max<int64_t>(n, 8 * cdiv(cdiv(s->H.DimUp, w), 8) * cdiv(s->H.Qdw * s->H.DimPh, std::max(R, 1)));
    }
    if (s->tr_on) n = std::max<int64_t>(n, cdiv(s->H.DimUp, 32) * cdiv(s->H.Qdw, 32));      // k_tadd: one partial per 32x32 tile
    if (s->csr_bpr) n = std::max<int64_t>(n, (int64_t)s->csr_bpr * s->H.Qdw * s->H.DimPh);
    if (s->csrD_bpr) n = std::max<int64_t>(n, 8 * cdiv(s->csrD_bpr, 8) * s->H.Qdw * s->H.DimPh);
    return n;
}

static int create(const edgpu_desc *desc, S **out)
{
    CHK(require_device());
    warn_unknown_knobs();
    auto s = std::unique_ptr<S>(new S());
    std::string e = s->H.init(*desc);
    if (!e.empty()) return fail(EDGPU_EINVAL, e);
    if (desc->storage < EDGPU_STORE_KRON || desc->storage > EDGPU_STORE_DIRECT) return fail(EDGPU_EINVAL, "edgpu: bad storage");
    if (desc->device >= 0) HIPCHK(hipSetDevice(desc->device));
    HIPCHK(hipGetDevice(&s->device));
    CHK(g_streams.take(s->device, &s->stream));      // plain non-blocking streams, recycled across sectors (runtime.hpp)
    CHK(g_streams.take(s->device, &s->cstream));
    HIPCHK(hipEventCreate(&s->ev_a)); HIPCHK(hipEventCreate(&s->ev_b));
    HIPCHK(hipEventCreateWithFlags(&s->ev_gather, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming));
    SectorHost &H = s->H;
    s->comm.rank = desc->mpi_rank; s->comm.size = desc->mpi_size;
    s->two_factor = desc->ed_total_ud != 0;
    s->extra = H.DimPh > 1 || H.Jhflag; s->has_ph = H.DimPh > 1; s->has_nd = H.Jhflag;
    s->nloc_el = H.DimUp * H.Qdw; s->nloc = s->nloc_el * H.DimPh; s->vecdim = s->nloc;
    s->multi = desc->mpi_size > 1 || (desc->reserved & 1);   // reserved bit 0: run the exchange path even on one rank (tests)
    s->lz_fused_on = knob("EDGPU_LZ_FUSED") != 0; s->lz_wg_on = knob("EDGPU_LZ_WG") != 0; s->sr_on = knob("EDGPU_LZ_SR") != 0; s->dvd_fused_on = knob("EDGPU_DVD_FUSED") != 0;
    s->dbg = dbg_flags(); s->one_ar = knob("EDGPU_ONE_ALLREDUCE") != 0; s->up_pairs10 = knob("EDGPU_UP_PAIRS10") != 0; s->direct_cache = XKNOB("EDGPU_DIRECT_CACHE", 1) != 0; s->up_elem16 = XKNOB("EDGPU_UP_ELEM16", 1) != 0; s->pad_allowed = knob("EDGPU_PAD") != 0 && knob("EDGPU_ODD_PAIRS") != 0; s->tables_wave = XKNOB("EDGPU_TABLES_WAVE", 1) != 0; s->dw_lds_kb = std::min(XKNOB("EDGPU_DW_LDS_KB", 0), 150);
    // distributed-transpose dw pass: opt-in (EDGPU_DW_TRANSPOSE=1) until an 8-GPU run has shown it correct and faster than the
    // needed-rows exchange; at most 64 ranks (larger jobs use the needed-rows exchange)
    s->tr_on = s->multi && s->two_factor && !s->extra && desc->storage != EDGPU_STORE_CSR && knob("EDGPU_DW_TRANSPOSE") != 0 && H.DimUp >= desc->mpi_size && desc->mpi_size <= 64;
    for (int k = 0; k < 2 * H.Ns_Ud; k++) CHK(upload_factor(s.get(), k));
    s->composite = !s->two_factor && H.cfac.size() == 2 && knob("EDGPU_COMPOSITE") != 0;
    s->kron2 = s->two_factor || s->composite;
    if (s->composite) { CHK(upload_factor(s.get(), -1)); CHK(upload_factor(s.get(), -2)); s->upf = s->cdev[0].get(); s->dwf = s->cdev[1].get(); s->hup = &H.cfac[0]; s->hdw = &H.cfac[1]; }
    else if (s->two_factor) { s->upf = s->fac[0].get(); s->dwf = s->fac[1].get(); s->hup = &H.fac[0]; s->hdw = &H.fac[1]; }
    CHK(s->cross.upload(H.cross)); CHK(s->gup.upload(H.gph_up)); CHK(s->gdw.upload(H.gph_dw));
    {
        std::vector<double> bu((size_t)desc->Nfoo * desc->Nbath), bd((size_t)desc->Nfoo * desc->Nbath);
        for (int a = 1; a <= desc->Nfoo; a++) for (int k = 1; k <= desc->Nbath; k++) { bu[(a - 1) + desc->Nfoo * (k - 1)] = H.bdiag(1, a, k); bd[(a - 1) + desc->Nfoo * (k - 1)] = H.bdiag(desc->Nspin, a, k); }
        std::vector<int32_t> st((size_t)desc->Norb * desc->Nbath);
        for (int a = 1; a <= desc->Norb; a++) for (int k = 1; k <= desc->Nbath; k++) st[(a - 1) + desc->Norb * (k - 1)] = H.stride[a - 1][k - 1];
        CHK(s->bd_up.upload(bu)); CHK(s->bd_dw.upload(bd)); CHK(s->stride.upload(st));
    }
    // launch geometry of the fused two-factor kernel
    if (s->kron2) {
        size_t rowbytes = (size_t)H.DimUp * sizeof(double);
        s->use_lds = rowbytes + 128 <= 150 * 1024 && !(dbg_flags() & 4);
        if (XKNOB("EDGPU_KRON_THREADS", 0)) s->kron_threads_override = XKNOB("EDGPU_KRON_THREADS", 0);
        // threads of the one-workgroup-per-row product: a lane walks DimUp / threads elements one after the other, each with its dependent
        // dw gathers; 512 threads from 384 columns on (Ns=12: 25.2 -> 18.9 us per Lanczos step at N = 174 240, 31 -> 27 at N = 853 776;
        // 1 024 threads lose again on the largest of those sectors: profiles/r04/kron_threads_small_sectors.txt)
        s->kron_threads = H.DimUp >= 4096 ? 1024 : (H.DimUp >= 384 ? 512 : 256);
        if (s->kron_threads_override) s->kron_threads = s->kron_threads_override;
        s->kron_lds = (s->use_lds ? rowbytes : 0) + 24 * sizeof(double);
        // two-pass product (local pass + column-tiled dw pass, fused Lanczos iteration) from 12 MiB per vector on: measured per Lanczos step at
        // Ns=14, single pass -> two passes: 8 MB 35 -> 36 us (on the fly 42 -> 37), 16 MB 56 -> 48 (68 -> 48), 32 MB 96 -> 75 (110 -> 72),
        // 48 MB 133 -> 95 (150 -> 93); below, the single-pass kernel with its two-launch iteration wins (profiles/r04/kron_split_threshold.txt).
        // Until round 4 the switch sat at 48 MiB.
        { const int k = knob("EDGPU_KRON_SPLIT"); s->kron_split = k < 0 ? H.DimUp * H.DimDw * 8 > ((int64_t)12 << 20) : k != 0; }
        // tile width of the dw pass: 64 columns measured best at Ns=16 (slab 6.6 MB, half of the gathers hit
        // L2) and at Ns=18 (slab 25 MB: no re-use left, but narrower tiles lose more to short segments)
        s->dw_w = knob("EDGPU_DW_W");
        s->dw_variant = knob("EDGPU_DW_VARIANT");
        s->up_persistent = knob("EDGPU_UP_PERSISTENT") != 0;
        s->up_grid = knob("EDGPU_UP_GRID");
        s->up_pairs = knob("EDGPU_UP_PAIRS") != 0;
        s->row_split = std::max(1, std::min(32, knob("EDGPU_ROW_SPLIT")));
        s->up_big = knob("EDGPU_UP_BIG") != 0;
        s->up_halo = knob("EDGPU_UP_HALO") != 0;

        if (!s->use_lds && s->up_halo && !s->tr_on) CHK(build_up_segments(s.get()));
        s->nd_split = knob("EDGPU_ND_SPLIT") != 0;
        s->up_threads = knob("EDGPU_UP_THREADS") == 512 ? 512 : 1024;
        s->dw_rows = std::max(1, std::min(64, knob("EDGPU_DW_ROWS")));
        if (s->dw_w != 8 && s->dw_w != 16 && s->dw_w != 32 && s->dw_w != 64 && s->dw_w != 128 && s->dw_w != 256) s->dw_w = 32;
    }
    // operator storage
    int64_t nnz_fac = 0;
    if (s->kron2) {
        const Factor &up = *s->hup, &dw = *s->hdw;
        int64_t nd = dw.rowptr[H.dw0 + H.Qdw] - dw.rowptr[H.dw0];
        nnz_fac = (int64_t)up.cols.size() * H.Qdw + nd * H.DimUp;
    } else {
        for (int k = 0; k < 2 * H.Ns_Ud; k++) nnz_fac += (int64_t)H.fac[k].cols.size() * (s->nloc_el / H.fac[k].dim);
    }
    s->nnz_total = (s->nloc_el + nnz_fac) * H.DimPh;
    s->multi = desc->mpi_size > 1 || (desc->reserved & 1);   // reserved bit 0: run the exchange path even on one rank (tests)
    s->nhalo = H.DimDw;
    if (s->multi && !s->tr_on && s->kron2 && (!s->extra || (s->has_nd && !s->has_ph && s->nd_split)) && knob("EDGPU_HALO")) CHK(setup_halo(s.get()));
    // the vectorised dw pass takes the rank's own rows from the input vector: no own-row copy into the x-space per product
    s->x_own_in_v = s->halo && s->kron2 && desc->storage != EDGPU_STORE_CSR && !s->extra && !s->tr_on && H.DimPh == 1 && s->dw_variant >= 2 && H.DimUp % 2 == 0 && s->dw_w >= 4 && s->dwf->width <= 64;
    if (s->tr_on) {     // up-block of this rank in the transposed space (same rule as the dw split)
        const int64_t P = s->comm.size, r = s->comm.rank, q = H.DimUp / P, rem = H.DimUp % P;
        s->Qup = r < rem ? q + 1 : q; s->up0 = r < rem ? r * (q + 1) : r * q + rem;
        CHK(s->tA.alloc((size_t)s->nloc_el)); CHK(s->tB.alloc((size_t)(s->Qup * H.DimDw)));
        CHK(s->tVT.alloc((size_t)(s->Qup * H.DimDw))); CHK(s->tHVT.alloc((size_t)(s->Qup * H.DimDw)));
        s->tr_lds = (size_t)H.DimDw * 8 + 128 <= 150 * 1024 && !(dbg_flags() & 4);
        s->tr_chunks = std::max(1, std::min(32, knob("EDGPU_TR_CHUNKS")));
    }
    if (desc->storage != EDGPU_STORE_DIRECT) { CHK(build_diag(s.get())); s->bytes_stored = s->nloc_el * 8; }
    if (desc->storage == EDGPU_STORE_CSR) {
        CHK(build_csr(s.get()));
        s->bytes_stored = s->csr_nnz * 12 + (s->nloc_el + 1) * 8 * (s->csr_three ? 3 : (s->csr_split ? 2 : 1)) + s->nd_nnz * 13 + (s->nd_nr + 1) * 16;
        s->nnz_total += s->nd_nnz;
    }
    if (s->multi && !s->tr_on) {
        const size_t nx = (size_t)(xspace_ld(s.get()) * s->nhalo * H.DimPh);
        CHK(s->vfull.alloc(nx));
        if (xspace_ld(s.get()) > H.DimUp) HIPCHK(hipMemsetAsync(s->vfull.p, 0, nx * 8, s->stream));
    }
    s->axpy_nb = std::max(1, XKNOB("EDGPU_AXPY_NB", 512));
    s->vld = H.DimUp; s->vlen = s->nloc;
    s->pad_on = (fused_ok(s.get()) || pitch_ok(s.get())) && (H.DimUp % 16) != 0 && knob("EDGPU_PAD") != 0;
    if (s->pad_on) { s->vld = (H.DimUp + 15) & ~(int64_t)15; s->vlen = (s->nloc / H.DimUp) * s->vld; }
    s->npartial = max_partials(s.get());
    CHK(s->partial.alloc((size_t)s->npartial));
    CHK(s->partial2.alloc(1024));
    CHK(s->scal.alloc(SC_COUNT));
    HIPCHK(hipMemsetAsync(s->scal.p, 0, SC_COUNT * sizeof(double), s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    *out = s.release();
    return EDGPU_OK;
}

static void destroy(S *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    delete s;      // ~edgpu_sector_s releases the communicator, events, streams and (through DevBuf) device memory
}

}  // namespace edgpu
