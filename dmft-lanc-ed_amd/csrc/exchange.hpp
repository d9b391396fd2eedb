// exchange.hpp -- the per-product exchange between ranks (needed dw rows or full gather) and the scalar all-reduce
// Part of libedgpu.so (one translation unit: edgpu.hip includes these headers in order).
#pragma once

namespace edgpu {

// ---------------------------------------------------------------------------------------
// exchange step: gather the input vector of every rank (phonon block by phonon block)
// ---------------------------------------------------------------------------------------
static int halo_exchange(S *s, const double *v_local, hipStream_t st);
// Emulated communicator (edgpu_comm_create_emulated): ONE rank of P runs alone on its GPU with the partition, row lists, x-space,
// packing and stream structure of the P-rank job; what the peers would deliver is replaced by device-to-device copies of the same
// byte count out of the rank's own data, issued on the comm stream where the receives would be (the write side of a receive is
// HBM traffic of this rank; the read side of the copy is not, and stands in for the sends this rank serves).  The numbers that
// arrive are not the peers' -- finite, of the right size -- so such a run measures time and traffic, never a result.
static int emu_fill(double *dst, int64_t count, const double *src, int64_t srccap, hipStream_t st)
{
    if (srccap < 1) return EDGPU_OK;
    for (int64_t o = 0; o < count; o += srccap)
        HIPCHK(hipMemcpyAsync(dst + o, src, (size_t)std::min(srccap, count - o) * 8, hipMemcpyDeviceToDevice, st));
    return EDGPU_OK;
}
// all-gather into the global layout [iph][idw][iup] at dst (allgather_vector_MPI, ED_SETUP.f90:689-733)
static int gather_full(S *s, const double *v_local, double *dst, hipStream_t st)
{
    const SectorHost &H = s->H; Comm &c = s->comm;
    const int P = c.size;
    std::vector<int64_t> counts(P), displs(P);
    for (int r = 0; r < P; r++) {
        int64_t q = H.DimDw / P, rem = H.DimDw % P;
        int64_t qd = r < rem ? q + 1 : q, d0 = r < rem ? r * (q + 1) : r * q + rem;
        counts[r] = qd * H.DimUp; displs[r] = d0 * H.DimUp;
    }
    const int64_t nel_loc = s->nloc_el, nel = H.DimUp * H.DimDw;
    for (int iph = 0; iph < H.DimPh; iph++) {
        const double *send = v_local + (int64_t)iph * nel_loc;
        double *recv = dst + (int64_t)iph * nel;
        if (c.kind == Comm::RCCL) {
            HIPCHK(hipMemcpyAsync(recv + displs[c.rank], send, (size_t)nel_loc * 8, hipMemcpyDeviceToDevice, st));
            int rc = g_rccl.GroupStart();
            for (int r = 0; r < P && rc == 0; r++) {
                if (r == c.rank) continue;
                rc = g_rccl.Send(send, (size_t)nel_loc, kNcclDouble, r, c.nccl, st);
                if (rc == 0) rc = g_rccl.Recv(recv + displs[r], (size_t)counts[r], kNcclDouble, r, c.nccl, st);
            }
            int rc2 = g_rccl.GroupEnd();
            if (rc || rc2) return fail(EDGPU_ECOMM, std::string("edgpu: RCCL send/recv failed: ") + g_rccl.GetErrorString(rc ? rc : rc2));
        } else if (c.kind == Comm::CALLBACK) {
            HIPCHK(hipStreamSynchronize(st));
            if (c.ag(c.ctx, send, nel_loc, recv, counts.data(), displs.data(), P)) return fail(EDGPU_ECOMM, "edgpu: allgatherv callback failed");
        } else if (c.kind == Comm::EMULATED) {
            for (int r = 0; r < P; r++) CHK(emu_fill(recv + displs[r], counts[r], send, nel_loc, st));
        } else
            return fail(EDGPU_ECOMM, "edgpu: mpi_size>1 but no communicator attached (edgpu_comm_init_*)");
    }
    return EDGPU_OK;
}
// The rows the peers need, packed on the COMPUTE stream in front of the local pass (hxv): beside the local pass the pack kernel is
// starved of workgroup slots and takes 0.6 ms where it takes 0.12 ms alone (Ns=16 explicit CSR, one rank of eight), and everything
// on the comm stream waits behind it.  (A high-priority comm stream has the same effect on one process per GPU, but with several
// processes sharing one device -- the test layout -- it could starve another process's compute queue for good: a rank hung inside
// its product while its peers waited in the next collective; profiles/r04/README.md.)
static int halo_pack(S *s, const double *v_local, hipStream_t st)
{
    const SectorHost &H = s->H;
    int64_t nsend_rows = 0; for (int r = 0; r < s->comm.size; r++) nsend_rows += s->send_rows_cnt[r];
    if (nsend_rows <= 0) return EDGPU_OK;
    const int64_t rowlen = xspace_ld(s);
    PackArgs K{}; K.DimUp = (int32_t)H.DimUp; K.DimPh = H.DimPh; K.Qdw = H.Qdw; K.nrows = nsend_rows; K.list = s->send_list.p; K.v = v_local; K.out = s->sendbuf.p; K.ld = rowlen;
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)std::min<int64_t>(nsend_rows * H.DimPh, 8192)), dim3(256), 0, st, K);
    HIPCHK(hipGetLastError());
    return EDGPU_OK;
}
static int gather_vector(S *s, const double *v_local, hipStream_t st)
{
    if (s->halo) return halo_exchange(s, v_local, st);
    return gather_full(s, v_local, s->vfull.p, st);
}

// x = [own rows | needed rows of the other ranks]: pack what each peer needs, one send/recv pair per peer
static int halo_exchange(S *s, const double *v_local, hipStream_t st)
{
    const SectorHost &H = s->H; Comm &c = s->comm;
    const int P = c.size;
    int64_t nsend_rows = 0; for (int r = 0; r < P; r++) nsend_rows += s->send_rows_cnt[r];
    // rows travel with the pitch of the x-space they land in: DimUp, or the line-aligned pitch of the explicit CSR (v_local has it too)
    const int64_t rowlen = xspace_ld(s);
    if (nsend_rows > 0 && !s->packed_ahead) {
        PackArgs K{}; K.DimUp = (int32_t)H.DimUp; K.DimPh = H.DimPh; K.Qdw = H.Qdw; K.nrows = nsend_rows; K.list = s->send_list.p; K.v = v_local; K.out = s->sendbuf.p; K.ld = rowlen;
        hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)std::min<int64_t>(nsend_rows * H.DimPh, 8192)), dim3(256), 0, st, K);
        HIPCHK(hipGetLastError());
    }
    if (s->relay_on) {      // two-phase relay routing (relay_plan, sector_dev.hpp); DimPh == 1
        double *x = s->vfull.p, *xr = x + H.Qdw * rowlen, *stg = s->relay_stage.p; const double *sb = s->sendbuf.p;
        if (!s->csr_three && !s->x_own_in_v) HIPCHK(hipMemcpyAsync(x, v_local, (size_t)(H.Qdw * rowlen) * 8, hipMemcpyDeviceToDevice, st));
        for (int phase = 1; phase <= 2; phase++) {
            // per peer: the direct piece and the relayed piece of this phase, each with the buffer it leaves from / lands in
            auto pieces = [&](int b, const S::RelayPiece *&sd, const S::RelayPiece *&sr, const S::RelayPiece *&rd, const S::RelayPiece *&rr) {
                const S::RelayPeer &Q = s->relay[b];
                if (phase == 1) { sd = &Q.s1d; sr = &Q.s1r; rd = &Q.r1d; rr = &Q.r1r; } else { sd = &Q.s2d; sr = &Q.s2f; rd = &Q.r2d; rr = &Q.r2f; }
            };
            const double *srsrc = phase == 1 ? sb : stg;      // relayed rows leave from my send buffer (phase 1) or from what I hold (phase 2)
            double *rrdst = phase == 1 ? stg : xr;            // and land in the holding buffer (phase 1) or in the x-space (phase 2)
            if (c.kind == Comm::RCCL) {
                int rc = g_rccl.GroupStart();
                for (int b = 0; b < P && rc == 0; b++) {
                    if (b == c.rank) continue;
                    const S::RelayPiece *sd, *sr, *rd, *rr; pieces(b, sd, sr, rd, rr);
                    // same order on both sides of a link: direct piece first, relayed piece second
                    if (sd->cnt) rc = g_rccl.Send(sb + sd->off * rowlen, (size_t)(sd->cnt * rowlen), kNcclDouble, b, c.nccl, st);
                    if (rc == 0 && sr->cnt) rc = g_rccl.Send(srsrc + sr->off * rowlen, (size_t)(sr->cnt * rowlen), kNcclDouble, b, c.nccl, st);
                    if (rc == 0 && rd->cnt) rc = g_rccl.Recv(xr + rd->off * rowlen, (size_t)(rd->cnt * rowlen), kNcclDouble, b, c.nccl, st);
                    if (rc == 0 && rr->cnt) rc = g_rccl.Recv(rrdst + rr->off * rowlen, (size_t)(rr->cnt * rowlen), kNcclDouble, b, c.nccl, st);
                }
                int rc2 = g_rccl.GroupEnd();
                if (rc || rc2) return fail(EDGPU_ECOMM, std::string("edgpu: RCCL send/recv failed: ") + g_rccl.GetErrorString(rc ? rc : rc2));
            } else if (c.kind == Comm::CALLBACK) {
                if (!c.xch) return fail(EDGPU_ECOMM, "edgpu: the callback exchange needs an all-to-all-v callback");
                std::vector<int64_t> sc(P, 0), sd_(P, 0), rcv(P, 0), rd_(P, 0);
                HIPCHK(hipStreamSynchronize(st));
                for (int kind = 0; kind < 2; kind++) {      // one all-to-all-v per kind of piece (a callback moves one piece per peer)
                    for (int b = 0; b < P; b++) {
                        sc[b] = sd_[b] = rcv[b] = rd_[b] = 0;
                        if (b == c.rank) continue;
                        const S::RelayPiece *sd, *sr, *rd, *rr; pieces(b, sd, sr, rd, rr);
                        const S::RelayPiece *ps = kind ? sr : sd, *pr = kind ? rr : rd;
                        sc[b] = ps->cnt * rowlen; sd_[b] = ps->off * rowlen; rcv[b] = pr->cnt * rowlen; rd_[b] = pr->off * rowlen;
                    }
                    if (c.xch(c.ctx, kind ? srsrc : sb, sc.data(), sd_.data(), kind ? rrdst : xr, rcv.data(), rd_.data(), P)) return fail(EDGPU_ECOMM, "edgpu: exchange callback failed");
                }
            } else if (c.kind == Comm::EMULATED) {
                for (int b = 0; b < P; b++) {
                    if (b == c.rank) continue;
                    const S::RelayPiece *sd, *sr, *rd, *rr; pieces(b, sd, sr, rd, rr);
                    CHK(emu_fill(xr + rd->off * rowlen, rd->cnt * rowlen, v_local, H.Qdw * rowlen, st));
                    CHK(emu_fill(rrdst + rr->off * rowlen, rr->cnt * rowlen, v_local, H.Qdw * rowlen, st));
                }
            } else
                return fail(EDGPU_ECOMM, "edgpu: mpi_size>1 but no communicator attached (edgpu_comm_init_*)");
        }
        return EDGPU_OK;
    }
    for (int iph = 0; iph < H.DimPh; iph++) {
        double *x = s->vfull.p + (int64_t)iph * s->nhalo * rowlen;
        // own rows of the x-space: not needed by the three-block CSR (its received-row block never reads them)
        if (!s->csr_three && !s->x_own_in_v) HIPCHK(hipMemcpyAsync(x, v_local + (int64_t)iph * s->nloc_el, (size_t)(H.Qdw * rowlen) * 8, hipMemcpyDeviceToDevice, st));      // DimPh > 1: rowlen == DimUp
        const double *sb = s->sendbuf.p + (int64_t)iph * nsend_rows * rowlen;
        if (P == 1) continue;
        if (c.kind == Comm::RCCL) {
            int rc = g_rccl.GroupStart();
            for (int r = 0; r < P && rc == 0; r++) {
                if (r == c.rank) continue;
                if (s->send_rows_cnt[r]) rc = g_rccl.Send(sb + s->send_rows_off[r] * rowlen, (size_t)(s->send_rows_cnt[r] * rowlen), kNcclDouble, r, c.nccl, st);
                if (rc == 0 && s->recv_rows_cnt[r]) rc = g_rccl.Recv(x + (H.Qdw + s->recv_rows_off[r]) * rowlen, (size_t)(s->recv_rows_cnt[r] * rowlen), kNcclDouble, r, c.nccl, st);
            }
            int rc2 = g_rccl.GroupEnd();
            if (rc || rc2) return fail(EDGPU_ECOMM, std::string("edgpu: RCCL send/recv failed: ") + g_rccl.GetErrorString(rc ? rc : rc2));
        } else if (c.kind == Comm::CALLBACK) {
            if (!c.xch) return fail(EDGPU_ECOMM, "edgpu: the callback exchange needs an all-to-all-v callback");
            std::vector<int64_t> sc(P), sd(P), rcv(P), rd(P);
            for (int r = 0; r < P; r++) { sc[r] = s->send_rows_cnt[r] * rowlen; sd[r] = s->send_rows_off[r] * rowlen; rcv[r] = s->recv_rows_cnt[r] * rowlen; rd[r] = (H.Qdw + s->recv_rows_off[r]) * rowlen; }
            HIPCHK(hipStreamSynchronize(st));
            if (c.xch(c.ctx, sb, sc.data(), sd.data(), x, rcv.data(), rd.data(), P)) return fail(EDGPU_ECOMM, "edgpu: exchange callback failed");
        } else if (c.kind == Comm::EMULATED) {
            for (int r = 0; r < P; r++)
                if (r != c.rank) CHK(emu_fill(x + (H.Qdw + s->recv_rows_off[r]) * rowlen, s->recv_rows_cnt[r] * rowlen, v_local + (int64_t)iph * s->nloc_el, H.Qdw * rowlen, st));
        } else
            return fail(EDGPU_ECOMM, "edgpu: mpi_size>1 but no communicator attached (edgpu_comm_init_*)");
    }
    return EDGPU_OK;
}

// sum n device doubles over ranks (in place)
static int allreduce_dev(S *s, double *p, int n)
{
    Comm &c = s->comm;
    if (!s->multi) return EDGPU_OK;
    if (c.kind == Comm::RCCL) {
        int rc = g_rccl.AllReduce(p, p, (size_t)n, kNcclDouble, kNcclSum, c.nccl, s->stream);
        if (rc) return fail(EDGPU_ECOMM, std::string("edgpu: RCCL allreduce failed: ") + g_rccl.GetErrorString(rc));
    } else if (c.kind == Comm::CALLBACK) {
        std::vector<double> h(n);
        HIPCHK(hipMemcpyAsync(h.data(), p, (size_t)n * 8, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
        if (c.ar(c.ctx, h.data(), n)) return fail(EDGPU_ECOMM, "edgpu: allreduce callback failed");
        HIPCHK(hipMemcpyAsync(p, h.data(), (size_t)n * 8, hipMemcpyHostToDevice, s->stream)); HIPCHK(hipStreamSynchronize(s->stream));
    } else if (c.kind == Comm::EMULATED) {
        // the rank's own partial sums stand for the global ones
    } else
        return fail(EDGPU_ECOMM, "edgpu: mpi_size>1 but no communicator attached");
    return EDGPU_OK;
}
// sum scal[SC_TMP] over ranks
static int allreduce_tmp(S *s) { return allreduce_dev(s, s->scal.p + SC_TMP, 1); }

}  // namespace edgpu
