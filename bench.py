#!/usr/bin/env python3
"""bench.py -- Lanczos HxV throughput of the Ns=16 half-filled sector on N MI355X.

A "step" is one full Lanczos iteration on the device-resident sector operator: the product
w = H v (fused with the beta*v_prev subtraction and the alpha dot), the axpy+norm pass and the
two single-workgroup reductions.  Default workload = BASELINE config 3: 1 orbital, Nbath=15
(Ns=16), Nup=Ndw=8, N=165,636,900, operator held as an explicit CSR in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` = equivalent-CSR bytes of one product (B_csr of the whole
sector) divided by the wall time of one Lanczos step (max over ranks): GB/s of HxV
sustained inside the Lanczos loop.  N>1 is strong scaling (same sector, dw columns split over
the ranks, RCCL exchange of the input vector every product).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))

HBM_PEAK_GBPS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def b_csr(N, nnz):
    """explicit CSR algorithmic bytes of one product: 12*nnz + 8*(N+1) + 16*N  """
    return 12 * nnz + 8 * (N + 1) + 16 * N


def cpu_baseline(seconds_budget=16.0, nbath=13):
    """The CPU oracle (a port: C restatement of spMatVec_main) on a bounded sample of the same workload
    family: the Ns=14 half-filled sector (N=11,778,624).  Timed twice: one thread (the reference has no
    threading) and with its row loops split over the host cores the process may use (OpenMP; bit-identical
    result).  `value` is the faster of the two, `cores` the threads that run used."""
    import numpy as np
    from edgpu import model
    from oracle import oracle
    m, nu, nd = model.half_filled_c(nbath)
    S = oracle.Sector(m, nu, nd)
    t0 = time.perf_counter(); S.buildh(); tb = time.perf_counter() - t0
    v = np.random.RandomState(0).uniform(-0.5, 0.5, S.Dim)
    N = S.Dim; nnz = N * (1 + m["Ns"])

    def run(threads, budget):
        oracle.set_threads(threads)
        S.spmatvec(v)
        n, t0 = 0, time.perf_counter()
        while True:
            S.spmatvec(v); n += 1
            dt = time.perf_counter() - t0
            if dt > budget or n >= 400:
                break
        return dt / n, n, dt
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    ncpu = max(1, min(ncpu, 16))        # a one-GPU box grants 16 cores' worth of CPU time whatever the affinity mask says
    t1, n1, d1 = run(1, seconds_budget / 2)
    tn, nn, dn = run(ncpu, seconds_budget / 2) if ncpu > 1 else (t1, n1, d1)
    oracle.set_threads(1)
    best, cores = (tn, ncpu) if tn < t1 else (t1, 1)
    return {"value": b_csr(N, nnz) / best / 1e9, "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": "oracle restatement of spMatVec_main (reference's stored format), Ns=%d half-filled sector (N=%d): %d products in %.1f s on 1 thread, %d products in %.1f s on %d threads (+%.1f s build); equivalent-CSR bytes / time"
                      % (m["Ns"], N, n1, d1, nn, dn, ncpu, tb),
            "s_per_hxv": best, "single_thread_value": b_csr(N, nnz) / t1 / 1e9, "single_thread_s_per_hxv": t1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nbath", type=int, default=15, help="Ns = nbath+1 (15 -> BASELINE config 3)")
    ap.add_argument("--storage", default="csr", choices=["csr", "kron", "direct"])
    ap.add_argument("--no-extra", action="store_true", help="skip the kron/direct side measurements")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch"], help="N>1: communicator owned by libedgpu.so (rccl) or torch.distributed device callbacks")
    ap.add_argument("--force-dist", action="store_true", help="run the N>1 code path (process group, communicator, exchange) even with one rank (self-test)")
    a = ap.parse_args()

    # exactly ONE line on stdout: libraries that chat on fd 1 (RCCL prints a version banner at communicator set-up)
    # are sent to stderr for the duration of the run; the JSON goes to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libedgpu.so has no CPU path)")
    torch.cuda.set_device(local_rank)
    dist = None
    multi = world > 1 or a.force_dist
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import numpy as np
    import edgpu
    from edgpu import model

    m, nu, nd = model.half_filled_c(a.nbath)
    t_build = time.perf_counter()
    G = edgpu.HvSector(m, nu, nd, storage=a.storage, rank=rank, size=world, device=local_rank, force_exchange=a.force_dist)
    comm_used = "none"
    if multi:
        ok = torch.ones(1, device="cuda")
        if a.comm == "rccl":
            # libedgpu.so opens its own RCCL communicator (id broadcast by torch.distributed)
            try:
                uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
                if rank == 0:
                    uid.copy_(torch.frombuffer(bytearray(edgpu.rccl_unique_id()), dtype=torch.uint8))
                dist.broadcast(uid, 0)
                G.comm_init_rccl(bytes(uid.cpu().numpy().tobytes()))
            except Exception as e:  # keep the job alive: every rank must take the same branch
                print("rank %d: native RCCL communicator failed (%s)" % (rank, e), file=sys.stderr)
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if a.comm == "torch" or ok.item() == 0:
            from edgpu import exchange
            ex = exchange.TorchDeviceExchange(dist)
            G.comm_init_callbacks(ex.allgatherv, ex.exchange, ex.allreduce)
            comm_used = "torch.distributed(nccl) device callbacks"
        else:
            comm_used = "native RCCL communicator (ncclSend/ncclRecv group + ncclAllReduce)"
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build

    K, W = a.steps, a.warmup
    G.bench_begin(W + K + 2)
    G.bench_steps(W, timed=False)
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    G.bench_steps(K, timed=True)
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    hxv_ms, alphas, betas = G.bench_result(K, ncoef=W + K)
    comm_ms = G.bench_comm_ms(K) if multi else 0.0

    N = int(G.Dim); Ns = m["Ns"]
    nnz = N * (1 + Ns)                   # 1 diag + Ns/2 up + Ns/2 dw per row at half filling, all V != 0
    Nloc = int(G.vecDim); nnz_loc = int(G.info.nnz_total)
    ms_step = dt / K * 1e3
    bytes_fmt = {"csr": b_csr(Nloc, nnz_loc), "kron": 24 * Nloc, "direct": 16 * Nloc}[a.storage]
    out = {
        "metric": "Lanczos HxV GB/s (equivalent-CSR bytes of one product / Lanczos-iteration time), Ns=%d half-filled sector" % Ns,
        "value": b_csr(N, nnz) / (dt / K) / 1e9, "unit": "GB/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "iters_per_sec": K / dt,
        "config": {"workload": "1-orbital Anderson impurity, Nbath=%d (Ns=%d), sector Nup=Ndw=%d, N=%d, default initial bath, operator storage=%s, one Lanczos iteration per step"
                               % (a.nbath, Ns, nu[0], N, a.storage),
                   "storage": a.storage, "N": N, "nnz": nnz, "parallelism": ("dw-column split x%d, needed-rows exchange per product over %s" % (world, comm_used)) if multi else "single GPU"},
        "roofline": {"bound": "hbm", "kernel": {"csr": "k_csr_spmv2<64,640,false> (local block) + k_csr_spmv2<64,640,true> (dw block): one product of the split explicit CSR",
                                "kron": "k_uppass + k_dwpass2 (two-pass stored product)", "direct": "k_direct_tables + k_uppass + k_dwpass2"}[a.storage],
                     "achieved": bytes_fmt / (hxv_ms * 1e-3) / 1e9 if hxv_ms > 0 else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": (bytes_fmt / (hxv_ms * 1e-3) / 1e9) / HBM_PEAK_GBPS if hxv_ms > 0 else None,
                     "algorithmic_bytes_per_launch": bytes_fmt, "kernel_ms": hxv_ms, "traffic": None},
        "time_split_ms": dict({"hxv_product": hxv_ms, "axpy_norm_reductions_scalars": ms_step - hxv_ms},
                              **({"exchange_on_comm_stream_overlapped_with_local_blocks": comm_ms} if multi else {})),
        "build_s": t_build,
        "lanczos_alpha_first_last": [float(alphas[0]), float(alphas[-1])] if len(alphas) else None,
    }
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            for rec in json.load(open(pmc)):
                if rec.get("storage") == a.storage and rec.get("nbath") == a.nbath and rec.get("n_gpus") == world:
                    out["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = rec.get("source")
        except Exception:
            pass
    if world == 1 and rank == 0 and not a.no_extra:
        extra = {}
        G.close()
        # measured streaming ceiling of this device: a plain device-to-device copy of two 1.3 GB vectors
        src = torch.empty(N, dtype=torch.float64, device="cuda").uniform_(); dst = torch.empty_like(src)
        for _ in range(3): dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): dst.copy_(src)
        e1.record(); torch.cuda.synchronize()
        out["roofline"]["measured_copy_GBps"] = 16.0 * N / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e9
        del src, dst
        for st in ("kron", "direct"):
            if st == a.storage:
                continue
            g = edgpu.HvSector(m, nu, nd, storage=st, device=local_rank)
            g.bench_begin(W + K + 2); g.bench_steps(W); torch.cuda.synchronize()
            t0 = time.perf_counter(); g.bench_steps(K, timed=True); torch.cuda.synchronize(); d2 = time.perf_counter() - t0
            ms, _, _ = g.bench_result(K)
            fb = {"kron": 24 * N, "direct": 16 * N}[st]
            extra[st] = {"ms_per_step": d2 / K * 1e3, "iters_per_sec": K / d2, "hxv_kernel_ms": ms,
                         "format_GBps": fb / (ms * 1e-3) / 1e9, "equivalent_csr_GBps": b_csr(N, nnz) / (ms * 1e-3) / 1e9}
            if st == "kron":      # the whole ground-state search the iterations serve (device-resident vector, second call: warm cache)
                dv = edgpu.DeviceVector(N)
                for _ in range(2):
                    dv.upload(np.zeros(N))
                    t0 = time.perf_counter(); e0, nl = g.sp_lanc_eigh_device(dv, nitermax=300, threshold=1e-12); tg = time.perf_counter() - t0
                dv.free()
                out["ground_state"] = {"storage": "kron", "E0": e0, "nlanc": nl, "seconds": tg}
            g.close()
        out["other_formats"] = extra
    if world == 1 and rank == 0 and not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
