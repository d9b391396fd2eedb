"""Worker of the multi-process tests: one rank of a world_size-P job (gloo rendezvous on
127.0.0.1).  mode=cpu : host logic only (partition + staged exchange + oracle slices).
mode=gpu : every rank drives libedgpu.so on the one visible GPU with the callback exchange."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    mode, case, outdir = sys.argv[1], sys.argv[2], sys.argv[3]
    import faulthandler
    faulthandler.dump_traceback_later(110, exit=False)      # a rank that is stuck says where (the launcher gives up after 150 s)
    import torch.distributed as dist
    rank, size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"], rank=rank, world_size=size)
    import cases
    from edgpu import model, exchange
    from oracle import oracle
    name, m, nu, nd = [z for z in cases.zoo() if z[0] == case][0]
    S = oracle.Sector(m, nu, nd)
    DimUp, DimDw, DimPh = S.DimUp, S.DimDw, S.DimPh
    q, d0 = model.dw_partition(DimDw, size, rank)
    nel_loc = DimUp * q
    rng = np.random.RandomState(123)
    vglob = rng.uniform(-0.5, 0.5, S.Dim)          # same on every rank
    ref = S.spmatvec(vglob)

    def local_slice(x):                             # [iph][idw_local][iup] layout of the MPI matvecs
        X = x.reshape(DimPh, DimDw, DimUp)
        return np.ascontiguousarray(X[:, d0:d0 + q, :]).reshape(-1)

    vloc = local_slice(vglob)
    res = {"rank": rank}
    if mode == "cpu":
        # staged allgatherv rebuilds the global vector block by block exactly as gather_vector() asks
        ex = exchange.StagedExchange(dist)
        counts = [model.dw_partition(DimDw, size, r)[0] * DimUp for r in range(size)]
        displs = [model.dw_partition(DimDw, size, r)[1] * DimUp for r in range(size)]
        full = np.zeros(S.Dim)
        for iph in range(DimPh):
            send = np.ascontiguousarray(vloc[iph * nel_loc:(iph + 1) * nel_loc])
            recv = full[iph * DimUp * DimDw:]
            ex.allgatherv(send.ctypes.data, send.size, recv.ctypes.data, counts, displs)
        assert np.array_equal(full, vglob)
        # all-to-all-v of the "needed rows only" exchange: rank r sends (r+1)*(q+1) doubles tagged (r,q) to rank q
        sc = [(rank + 1) * (r + 1) if r != rank else 0 for r in range(size)]
        sd = list(np.cumsum([0] + sc[:-1]))
        rc = [(r + 1) * (rank + 1) if r != rank else 0 for r in range(size)]
        rd = list(np.cumsum([0] + rc[:-1]))
        sbuf = np.concatenate([np.full(c, 100.0 * rank + r) for r, c in enumerate(sc)] + [np.zeros(1)])
        rbuf = np.full(sum(rc) + 1, -1.0)
        ex.exchange(sbuf.ctypes.data, sc, [int(x) for x in sd], rbuf.ctypes.data, rc, [int(x) for x in rd])
        for r in range(size):
            if r != rank:
                assert np.all(rbuf[rd[r]:rd[r] + rc[r]] == 100.0 * r + rank)
        part = np.array([float(vloc @ local_slice(ref))])
        ex.allreduce(part)
        assert abs(part[0] - vglob @ ref) <= 1e-10 * abs(vglob @ ref) + 1e-12
        res["ok"] = True
    elif mode == "cpu_halo":
        # the needed-rows exchange (csrc/sector_dev.hpp: setup_halo) restated on the oracle's H_dw: every rank derives BOTH
        # its receive lists and what every peer needs from it without communication; the lists must match pairwise and the
        # exchanged rows must be exactly the dw rows the rank's own rows couple to
        ex = exchange.StagedExchange(dist)
        rp, cols, vals = S.sp("dw")
        nbr = [set(int(c) - 1 for c in cols[rp[i]:rp[i + 1]]) for i in range(DimDw)]
        own = lambda r: range(model.dw_partition(DimDw, size, r)[1], model.dw_partition(DimDw, size, r)[1] + model.dw_partition(DimDw, size, r)[0])
        owner = {}
        for r in range(size):
            for j in own(r):
                owner[j] = r
        need = lambda r: sorted(set().union(*[nbr[i] for i in own(r)]) - set(own(r))) if len(own(r)) else []
        recv_rows = [[j for j in need(rank) if owner[j] == r] for r in range(size)]
        send_rows = [[j for j in need(r) if owner[j] == rank] for r in range(size)]
        X = vglob.reshape(DimPh, DimDw, DimUp)[0]
        sbuf = np.concatenate([X[j] for r in range(size) for j in send_rows[r]] + [np.zeros(1)])
        sc = [len(send_rows[r]) * DimUp for r in range(size)]; sd = [int(x) for x in np.cumsum([0] + sc[:-1])]
        rc = [len(recv_rows[r]) * DimUp for r in range(size)]; rd = [int(x) for x in np.cumsum([0] + rc[:-1])]
        rbuf = np.full(sum(rc) + 1, np.nan)
        ex.exchange(sbuf.ctypes.data, sc, sd, rbuf.ctypes.data, rc, rd)
        k = 0
        for r in range(size):
            for j in recv_rows[r]:
                assert np.array_equal(rbuf[k:k + DimUp], X[j]); k += DimUp
        res["ok"] = True; res["halo_rows"] = sum(len(x) for x in recv_rows); res["allgather_rows"] = DimDw - q
    elif mode == "cpu_transpose":
        # the distributed-transpose scheme of the dw hops (csrc/product.hpp: transposed_dw_pass) restated with numpy on
        # the same block arithmetic (counts / displacements of both all-to-alls), through the staged gloo exchange
        ex = exchange.StagedExchange(dist)
        P = size
        split = lambda D, r: model.dw_partition(D, P, r)
        qup, up0 = split(DimUp, rank)
        rng2 = np.random.RandomState(7)
        Hdw = rng2.uniform(-1, 1, (DimDw, DimDw)); Hdw = Hdw + Hdw.T             # any symmetric dw factor
        want = (Hdw @ vglob.reshape(DimPh, DimDw, DimUp)[0])[d0:d0 + q, :]        # (H_dw x 1) v, my dw rows
        vl = vloc[:nel_loc].reshape(q, DimUp)
        tA = np.ascontiguousarray(vl.T).reshape(-1)                              # DimUp x Qdw
        sc = [split(DimUp, r)[0] * q for r in range(P)]; sd = [q * split(DimUp, r)[1] for r in range(P)]
        rc = [qup * split(DimDw, r)[0] for r in range(P)]; rd = [qup * split(DimDw, r)[1] for r in range(P)]
        tB = np.zeros(qup * DimDw + 1)
        tB[rd[rank]:rd[rank] + rc[rank]] = tA[sd[rank]:sd[rank] + sc[rank]]
        s2, r2 = list(sc), list(rc); s2[rank] = 0; r2[rank] = 0
        ex.exchange(tA.ctypes.data, s2, sd, tB.ctypes.data, r2, rd)
        vt = np.zeros((qup, DimDw))
        for s_ in range(P):
            qs, ds = split(DimDw, s_)
            vt[:, ds:ds + qs] = tB[rd[s_]:rd[s_] + rc[s_]].reshape(qup, qs)
        hvt = vt @ Hdw.T                                                         # row pass: hvt(il, i) = sum_j Hdw[i,j] vt(il, j)
        back = np.zeros(qup * DimDw + 1)
        for s_ in range(P):
            qs, ds = split(DimDw, s_)
            back[rd[s_]:rd[s_] + rc[s_]] = np.ascontiguousarray(hvt[:, ds:ds + qs]).reshape(-1)
        tA2 = np.zeros(DimUp * q + 1)
        tA2[sd[rank]:sd[rank] + sc[rank]] = back[rd[rank]:rd[rank] + rc[rank]]
        ex.exchange(back.ctypes.data, r2, rd, tA2.ctypes.data, s2, sd)
        got = tA2[:DimUp * q].reshape(DimUp, q).T
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
        res["ok"] = True
    elif mode == "sweep":
        # one communicator object per rank for the whole sweep (callback form: the host owns the groups, so the shrunk
        # communicator of a sector with DimDw < size is a callback communicator over dist.new_group(first DimDw ranks))
        import edgpu
        ex = exchange.device_staged(dist)
        world = edgpu.Communicator.callbacks(rank, size, ex.allgatherv, ex.exchange, ex.allreduce)
        groups = {n: dist.new_group(list(range(n))) for n in range(1, size)}          # collective: every rank creates every group
        subs = {}
        Nso = m["Ns"]
        nsec = shrunk = skipped = 0
        for nup in range(Nso + 1):
            for ndw in range(Nso + 1):
                St = oracle.Sector(m, [nup], [ndw])
                wd = np.linalg.eigvalsh(St.hmat())
                nsec += 1
                P = min(size, St.DimDw)                                               # ED_HAMILTONIAN.f90:65-93
                if P < size:
                    shrunk += 1
                    if rank >= P:
                        skipped += 1
                        continue
                    if P not in subs:
                        exs = exchange.device_staged(dist, group=groups[P])
                        subs[P] = edgpu.Communicator.callbacks(rank, P, exs.allgatherv, exs.exchange, exs.allreduce)
                    c = subs[P]
                else:
                    c = world
                G = edgpu.HvSector(m, [nup], [ndw], storage="kron" if (nup + ndw) % 2 else "direct", rank=rank, size=P)
                G.attach_comm(c)
                e0, x, nl = G.sp_lanc_eigh(nitermax=min(300, St.Dim), threshold=1e-14)
                assert abs(e0 - wd[0]) < 1e-9, (nup, ndw, e0, wd[0])
                G.close()
        res.update(ok=True, nsectors=nsec, shrunk=shrunk, skipped=skipped)
    elif mode == "torchdev":
        import torch  # noqa: F401  (before libedgpu.so, as in bench.py)
        import edgpu
        ex = exchange.TorchDeviceExchange(dist)
        for m2 in (m, dict(m, Nph=1, w0_ph=0.3, g_ph=(0.2,))):      # halo path and full-gather path
            B = edgpu.HvSector(m2, nu, nd, storage="kron", force_exchange=True)
            B.comm_init_callbacks(ex.allgatherv, ex.exchange, ex.allreduce)
            x = cases.sinvec(B.Dim)
            A = edgpu.HvSector(m2, nu, nd, storage="kron")
            r0 = A.spHtimesV(x)
            assert np.abs(B.spHtimesV(x) - r0).max() <= 1e-13 * np.abs(r0).max()
            e1, _, _ = B.sp_lanc_eigh(x, nitermax=100, threshold=1e-13)
            e2, _, _ = A.sp_lanc_eigh(x, nitermax=100, threshold=1e-13)
            assert abs(e1 - e2) < 1e-11
            A.close(); B.close()
        res["ok"] = True
    elif mode == "legs":
        # multi-rank forms of the callers either side of the product (GF / chi / phonon legs, observables,
        # twin hand-off) against the single-rank library on the same GPU (itself checked against the oracle)
        import edgpu
        ex = exchange.device_staged(dist)
        total_ud = m.get("ed_total_ud", True)
        w, U = np.linalg.eigh(S.hmat()); gs = U[:, 0]
        gloc = local_slice(gs)
        G = edgpu.HvSector(m, nu, nd, storage="kron", rank=rank, size=size)
        G.comm_init_callbacks(ex.allgatherv, ex.exchange, ex.allreduce)
        G1 = edgpu.HvSector(m, nu, nd, storage="kron")
        np.testing.assert_allclose(G.imp_histogram(gloc), G1.imp_histogram(gs), rtol=0, atol=1e-13)
        o, o1 = G.observables(gloc), G1.observables(gs)
        for k in o1:
            np.testing.assert_allclose(o[k], o1[k], rtol=0, atol=1e-12)
        if total_ud and m["Norb"] > 1:
            np.testing.assert_allclose(G.imp_density_offdiag(gloc), G1.imp_density_offdiag(gs), rtol=0, atol=1e-13)
        le, le1 = G.local_energy(gloc), G1.local_energy(gs)      # Dse / Dph partner rows live on other ranks
        lo = oracle.local_energy(S, gs)
        for k in le1:
            assert abs(le[k] - le1[k]) <= 1e-12 * max(1.0, abs(le1[k])) and abs(le1[k] - lo[k]) <= 1e-12 * max(1.0, abs(lo[k])), (k, le[k], le1[k], lo[k])
        nl = min(S.Dim, 30)
        cu = [0.5] + [0.0] * (m["Norb"] - 1); cd = [-0.5] + [0.25] * (m["Norb"] - 1)
        r, r1 = G.chi_lanczos(cu, cd, gloc, nl), G1.chi_lanczos(cu, cd, gs, nl)
        assert abs(r[0] - r1[0]) < 1e-13 and r[3] == r1[3]
        np.testing.assert_allclose(r[1][:20], r1[1][:20], rtol=1e-9, atol=1e-9); np.testing.assert_allclose(r[2][1:20], r1[2][1:20], rtol=1e-9, atol=1e-9)
        if DimPh > 1:
            r, r1 = G.phonon_lanczos(gloc, nl), G1.phonon_lanczos(gs, nl)
            assert abs(r[0] - r1[0]) < 1e-13
            np.testing.assert_allclose(r[1][:20], r1[1][:20], rtol=1e-9, atol=1e-9); np.testing.assert_allclose(r[2][1:20], r1[2][1:20], rtol=1e-9, atol=1e-9)
        # twin hand-off: shard of the twin sector B(ndw,nup)
        tw1 = G1.twin_reorder(gs).reshape(DimPh, DimUp, DimDw)      # B: DimUp_B = DimDw, DimDw_B = DimUp
        qb, b0 = model.dw_partition(DimUp, size, rank)
        assert np.array_equal(G.twin_reorder(gloc), np.ascontiguousarray(tw1[:, b0:b0 + qb, :]).reshape(-1))
        nlegs = 0
        for iorb in range(1, m["Norb"] + 1):
            ia = 0 if total_ud else iorb - 1
            for ispin in (1, 2):
                for dagger in (True, False):
                    nu2, nd2 = list(nu), list(nd)
                    (nu2 if ispin == 1 else nd2)[ia] += 1 if dagger else -1
                    nlev = m["Ns"] if total_ud else 1 + m["Nbath"]
                    if min(nu2 + nd2) < 0 or max(nu2 + nd2) > nlev:
                        continue
                    St = oracle.Sector(m, nu2, nd2)
                    if St.DimDw < size:
                        continue
                    Gt = edgpu.HvSector(m, nu2, nd2, storage="kron", rank=rank, size=size)
                    Gt.comm_init_callbacks(ex.allgatherv, ex.exchange, ex.allreduce)
                    Gt1 = edgpu.HvSector(m, nu2, nd2, storage="kron")
                    vref, n2ref = oracle.apply_op(S, St, iorb, ispin, dagger, gs)
                    n = min(St.Dim, 25)
                    n2, a, b, used = edgpu.gf_lanczos(G, Gt, iorb, ispin, dagger, gloc, n)
                    n2s, a1, b1, used1 = edgpu.gf_lanczos(G1, Gt1, iorb, ispin, dagger, gs, n)
                    assert abs(n2 - n2ref) <= 1e-13 and abs(n2s - n2ref) <= 1e-13, (iorb, ispin, dagger, n2, n2ref)
                    if n2ref > 1e-20:
                        k = min(used, used1, 20)
                        np.testing.assert_allclose(a[:k], a1[:k], rtol=1e-9, atol=1e-9)
                        np.testing.assert_allclose(b[1:k], b1[1:k], rtol=1e-9, atol=1e-9)
                        # the operator-applied state itself, shard by shard
                        qt, t0 = model.dw_partition(St.DimDw, size, rank)
                        din = edgpu.DeviceVector(gloc.size).upload(gloc); dout = edgpu.DeviceVector(Gt.vecDim)
                        edgpu.apply_op(G, Gt, iorb, ispin, dagger, din, dout)
                        want = np.ascontiguousarray(vref.reshape(DimPh, St.DimDw, St.DimUp)[:, t0:t0 + qt, :]).reshape(-1)
                        np.testing.assert_allclose(dout.download(), want, rtol=0, atol=1e-14)
                        nlegs += 1
                    Gt.close(); Gt1.close()
        G.close(); G1.close()
        res["ok"] = True; res["nlegs"] = nlegs
    else:
        import edgpu
        csr_ok = True
        for st in (["kron", "direct", "csr"] if csr_ok else ["kron", "direct"]):
            G = edgpu.HvSector(m, nu, nd, storage=st, rank=rank, size=size)
            ex = exchange.device_staged(dist)
            G.comm_init_callbacks(ex.allgatherv, ex.exchange, ex.allreduce)
            assert G.vecDim == vloc.size and G.mpiQdw == q and G.mpiIshift == d0 * DimUp
            hv = G.spHtimesV(vloc)
            err = np.abs(hv - local_slice(ref)).max()
            assert err <= 1e-12 * np.abs(ref).max(), (st, err)
            if st == "kron" and S.Dim <= 1500:      # dense form, rows summed over the ranks: every rank holds the full matrix
                Href = S.hmat()
                assert np.abs(G.hmat() - Href).max() <= 1e-15 * max(1.0, np.abs(Href).max())
            v0 = local_slice(np.random.RandomState(5).uniform(0, 1, S.Dim))
            e0, x, nl = G.sp_lanc_eigh(v0, nitermax=min(300, S.Dim), threshold=1e-14)
            res["e0_" + st] = e0
            n0 = ex.n_allreduce
            a, b, used = G.sp_lanc_tridiag(v0, 20)
            res["a_" + st] = a.tolist(); res["b_" + st] = b.tolist()
            res["allreduce_calls_tridiag20_" + st] = ex.n_allreduce - n0
            if os.environ.get("EDGPU_RELAY"):
                ri = G.relay_info()
                res["relay_" + st] = dict(enabled=ri["enabled"], chunks=ri["chunks"], before=ri["busiest_before"], after=ri["busiest_after"],
                                          direct=ri["rows_direct"].tolist(), routed=ri["rows_routed"].tolist(), mode=G.exchange_info()[0])
            if 8 < S.Dim <= 500 and st == "kron":      # the multi-eigenpair solvers on the sharded operator (small cases: the staged host exchange is slow)
                wd = np.linalg.eigvalsh(S.hmat())[:2]
                evd, _, info = G.sp_dvdson_eigh(2, nitermax=400, tol=1e-11)
                assert info["nconv"] == 2 and np.abs(evd - wd).max() < 1e-9, (st, evd, wd)
                eva, _, info = G.sp_eigh(2, nblock=min(S.Dim, 20), nitermax=200, tol=1e-12)
                assert info["nconv"] == 2 and np.abs(eva - wd).max() < 1e-9, (st, eva, wd)
            # eigenvector: gather on the host and check the residual with the oracle
            outs = [None] * size
            dist.all_gather_object(outs, (d0, q, x))
            X = np.zeros((DimPh, DimDw, DimUp))
            for (dd, qq, xx) in outs:
                X[:, dd:dd + qq, :] = xx.reshape(DimPh, qq, DimUp)
            xg = X.reshape(-1)
            assert abs(np.linalg.norm(xg) - 1) < 1e-10
            assert np.linalg.norm(S.spmatvec(xg) - e0 * xg) < 1e-6
            G.close()
        res["ok"] = True
    import json
    json.dump(res, open(os.path.join(outdir, "rank%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
