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
    else:
        import edgpu
        for st in (["kron", "direct", "csr"] if (m.get("Nph", 0) == 0 and not (m["Norb"] > 1 and (m.get("Jx", 0) or m.get("Jp", 0)))) else ["kron", "direct"]):
            G = edgpu.HvSector(m, nu, nd, storage=st, rank=rank, size=size)
            ex = exchange.device_staged(dist)
            G.comm_init_callbacks(ex.allgatherv, ex.exchange, ex.allreduce)
            assert G.vecDim == vloc.size and G.mpiQdw == q and G.mpiIshift == d0 * DimUp
            hv = G.spHtimesV(vloc)
            err = np.abs(hv - local_slice(ref)).max()
            assert err <= 1e-12 * np.abs(ref).max(), (st, err)
            v0 = local_slice(np.random.RandomState(5).uniform(0, 1, S.Dim))
            e0, x, nl = G.sp_lanc_eigh(v0, nitermax=min(300, S.Dim), threshold=1e-14)
            res["e0_" + st] = e0
            a, b, used = G.sp_lanc_tridiag(v0, 20)
            res["a_" + st] = a.tolist(); res["b_" + st] = b.tolist()
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
