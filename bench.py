#!/usr/bin/env python3
"""bench.py -- Lanczos HxV throughput of the Ns=16 half-filled sector on 1/2/4/8 MI355X.

A "step" is one full Lanczos iteration on the device-resident sector operator: the product
w = H v (fused with the beta*v_prev subtraction and the alpha dot), the axpy+norm pass and the
two single-workgroup reductions.  Default workload = BASELINE config 3: 1 orbital, Nbath=15
(Ns=16), Nup=Ndw=8, N=165,636,900, operator held as an explicit CSR in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]        # N>1: bench.py starts its own N rank processes
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   # or an external launcher
  python bench.py --gpus 8 --nbath 17 --storage direct       # the Ns=18 run (BASELINE config 4, on-the-fly operator)

Rank 0 prints ONE JSON line.  `value` = equivalent-CSR bytes of one product (B_csr of the whole
sector) divided by the wall time of one Lanczos step (max over ranks): GB/s of HxV
sustained inside the Lanczos loop.  N>1 is strong scaling (same sector, dw columns split over
the ranks as ED_HAMILTONIAN.f90:96-110 does, RCCL exchange of the needed rows of the input vector every product).
If the world that comes up differs from --gpus the run exits non-zero instead of reporting fewer GPUs.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))

HBM_PEAK_GBPS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
KERNELS = {"csr": "k_csr_il<false,0> (local block) + k_csr_il<true,1> (dw block): one product of the split explicit CSR, 64-row blocks stored lane-major",
           "kron": "k_uppass2_ax + k_dwpass2 (fused Lanczos iteration over the stored factors: axpy+norm+local pass, dw pass+scale+dot)",
           "direct": "k_uppass2_ax<DIRECT> + k_dwpass2 (fused Lanczos iteration, on-the-fly operator; the hop tables are derived from the bit patterns once per sector, k_direct_tables_wave)"}


BOUND_NOTE = {"csr": "explicit CSR inside the Lanczos loop is bounded at 0.57-0.60 of the 8 TB/s peak on this device: the split form moves 43.1 GB per product "
                     "(37.8 algorithmic + 4.0 second read of x and read-modify-write of the result + 1.3 u_prev) and every read/write mix measured here runs at "
                     "5.3-6.1 TB/s depending on the box (pure reads 7.07): floor 7.1-7.4 ms, measured 7.9-8.3 ms (HIP events) / +3-4 % under rocprofv3; "
                     "DESIGN.md section 4.  Round 4: a fused iteration (axpy+norm pass inside an LDS-row form of the local block) was built and measured on four boxes: -3.4 %, +1 %, +5 % per step -- not shipped (profiles/r04/csr_fused_iteration.txt)"}


def b_csr(N, nnz):
    """explicit CSR algorithmic bytes of one product: 12*nnz + 8*(N+1) + 16*N  """
    return 12 * nnz + 8 * (N + 1) + 16 * N


PRODUCT_SOURCES = ("exchange.hpp", "kernels.hpp", "product.hpp", "sector_dev.hpp", "sector_host.hpp")


def source_stamp():
    """sha256 over the sources that define the product kernels, their launches and the storage forms (PRODUCT_SOURCES; the solver
    loops, the C-ABI file and the host runtime around them cannot change what one product moves): the PMC traffic summary under
    profiles/ is stamped with it, and bench.py only reports that traffic while the stamp still matches the code that is running
    (there is no .git on the GPU box)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "dmft-lanc-ed_amd", "csrc")
    for f in PRODUCT_SOURCES:
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


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

    def run(threads, budget, sec=None, vec=None):
        sec = S if sec is None else sec; vec = v if vec is None else vec
        oracle.set_threads(threads)
        sec.spmatvec(vec)
        n, t0 = 0, time.perf_counter()
        while True:
            sec.spmatvec(vec); n += 1
            dt = time.perf_counter() - t0
            if dt > budget or n >= 400:
                break
        return dt / n, n, dt
    try:
        naff = len(os.sched_getaffinity(0))
    except AttributeError:
        naff = os.cpu_count() or 1
    # BASELINE config 2 (Ns=12, N=853,776) on the port as well: the one sector for which a timing of the reference's OWN code
    # exists (0.272 s per product, survey probe) -- the ratio says how much faster than the reference this port is
    m2, nu2, nd2 = model.half_filled_c(11)
    S2 = oracle.Sector(m2, nu2, nd2); S2.buildh()
    v2 = np.random.RandomState(2).uniform(-0.5, 0.5, S2.Dim)
    t_c2, n_c2, d_c2 = run(1, 1.0, S2, v2)
    # threads: what the affinity mask allows is only an upper bound (a container may be granted less CPU time than it has
    # cores visible): a ~1 s scaling probe on the small sector picks the count that actually runs fastest
    probe = {}
    cand = sorted({c for c in (1, 2, 4, 8, 12, 16, 24, 32, 48, 64, naff) if c <= naff})
    for c in cand:
        probe[c] = run(c, 1.0 / max(len(cand), 1), S2, v2)[0]
    ncpu = min(probe, key=probe.get)
    del S2, v2
    t1, n1, d1 = run(1, seconds_budget / 2)
    tn, nn, dn = run(ncpu, seconds_budget / 2) if ncpu > 1 else (t1, n1, d1)
    oracle.set_threads(1)
    best, cores = (tn, ncpu) if tn < t1 else (t1, 1)
    # the bench sector itself (Ns=16) on the CPU: its stored matrix is out of reach of the port (45 GB and minutes of build),
    # so a random sample of its rows through the oracle's on-the-fly row evaluator, scaled to the whole product
    ns16 = None
    try:
        m16, nu16, nd16 = model.half_filled_c(15)
        S16 = oracle.Sector(m16, nu16, nd16)
        rs = np.random.RandomState(1)
        v16 = rs.uniform(-0.5, 0.5, S16.Dim)
        nrow = 200000
        idx = np.sort(rs.randint(0, S16.Dim, nrow))
        S16.rows(idx[:1000], v16)
        t0 = time.perf_counter(); S16.rows(idx, v16); dt = time.perf_counter() - t0
        est = dt / nrow * S16.Dim
        ns16 = {"rows": nrow, "seconds": dt, "est_s_per_hxv_1thread": est, "est_equivalent_csr_GBps_1thread": b_csr(S16.Dim, S16.Dim * 17) / est / 1e9,
                "what": "orc_rows (rows generated from the bit patterns, reference's direct/ arithmetic) on a random sample of rows of the Ns=16 bench sector, scaled by N/rows"}
        del v16, S16
    except MemoryError:
        pass
    return {"value": b_csr(N, nnz) / best / 1e9, "unit": "GB/s", "cores": cores, "kind": "port", "ns16_rows_sample": ns16,
            "value_is": "port (C restatement of the reference's stored product, NOT the reference's own code), %d thread(s), Ns=%d half-filled sector" % (cores, m["Ns"]),
            "threads_probe": {"affinity_mask_cores": naff, "s_per_c2_product_by_threads": {str(k): round(x, 6) for k, x in sorted(probe.items())}, "chosen": ncpu},
            "c2_port_vs_reference_code": {"port_s_per_hxv_1thread": t_c2, "reference_code_s_per_hxv_recorded": 0.272, "port_speedup_over_reference_code": 0.272 / t_c2,
                                          "note": "same sector (Ns=12, N=853,776), one thread each; the port keeps flat CSR arrays where the reference walks per-row allocatable arrays, so `value` OVER-states what the reference's own CPU path would reach"},
            "sample": "oracle restatement of spMatVec_main (reference's stored format), Ns=%d half-filled sector (N=%d): %d products in %.1f s on 1 thread, %d products in %.1f s on %d threads (+%.1f s build); equivalent-CSR bytes / time"
                      % (m["Ns"], N, n1, d1, nn, dn, ncpu, tb),
            "s_per_hxv": best, "single_thread_value": b_csr(N, nnz) / t1 / 1e9, "single_thread_s_per_hxv": t1,
            "reference_code_recorded": {"s_per_hxv": 0.272, "config": "Ns=12 half-filled sector (BASELINE config 2), stored spMatVec_main, 1 core of the survey container",
                                        "equivalent_csr_GBps": 0.56, "source": "BASELINE.md section 2 (survey probe of the reference's own code; not re-measured here: the reference is not buildable in this repository)"}}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def launch_ranks(a, argv):
    """`python bench.py --gpus N` with no launcher around it: start N rank processes (one per GPU) and relay rank 0's
    line.  The parent never touches the GPU (children are fork+exec'ed; a process that has initialised HIP must not exec)."""
    import torch                                           # device_count() does not initialise the GPU runtime
    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.stderr.write("bench.py needs a GPU (libedgpu.so has no CPU path)\n"); return 2
    if ndev < a.gpus and not a.share_gpus:
        sys.stderr.write("bench.py: --gpus %d but only %d device(s) visible; refusing to report fewer GPUs than asked "
                         "(--share-gpus maps several ranks onto one device for functional tests)\n" % (a.gpus, ndev))
        return 2
    port = _free_port()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % ndev), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    t0 = time.time(); failed = None
    while True:                                            # a rank that dies leaves the others inside a collective: stop them at once
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]; break
        if all(c == 0 for c in codes):
            break
        if time.time() - t0 > a.launch_timeout:
            failed = -1; break
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.kill()                                   # exact children only
    out0 = procs[0].communicate()[0]
    for p in procs[1:]:
        p.wait()
    if failed is not None:
        sys.stderr.write("bench.py: %s\n" % ("timed out after %d s" % a.launch_timeout if failed == -1 else "rank %d exited with code %s" % (failed, procs[failed].returncode)))
        return 3
    sys.stdout.write(out0.decode()); sys.stdout.flush()
    return 0


def timed_steps(G, K, W, torch, dist):
    """W untimed + K timed Lanczos steps, barrier + synchronize on both sides, max over ranks"""
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
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def emulate_rank(a):
    """One rank of `--of` P, alone on one GPU (ED_HAMILTONIAN.f90:96-110 partition, same row lists, x-space, pack kernel, streams as
    the P-rank job; Communicator.emulated).  Prints one JSON line: per-step and per-product times of THAT rank's share, the exchange
    volume it would receive, and the emulated exchange (device copies) timed on the comm stream.  `value` is the rank's share of the
    equivalent-CSR bytes per step time -- a per-rank figure for the model of DESIGN section 6, not a job throughput."""
    import torch
    import numpy as np
    import edgpu
    from edgpu import model
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libedgpu.so has no CPU path)")
    r, P = a.emulate_rank, a.of
    m, nu, nd = model.half_filled_c(a.nbath)
    t0 = time.perf_counter()
    G = edgpu.HvSector(m, nu, nd, storage=a.storage, rank=r, size=P, device=0)
    comm = edgpu.Communicator.emulated(r, P)
    G.attach_comm(comm)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    K, W = a.steps, a.warmup
    dt = timed_steps(G, K, W, torch, None)
    hxv_ms, _, _ = G.bench_result(K)
    comm_ms = G.bench_comm_ms(K)
    xmode, sendb, recvb = G.exchange_info()
    N = int(G.Dim); Ns = m["Ns"]; Nloc = int(G.vecDim); nnz_loc = int(G.info.nnz_total)
    bytes_fmt = {"csr": b_csr(Nloc, nnz_loc), "kron": 24 * Nloc, "direct": 16 * Nloc}[a.storage]
    out = {"metric": "EMULATED rank %d of %d on ONE GPU: Lanczos-iteration time of that rank's share, Ns=%d half-filled sector (not a job throughput)" % (r, P, Ns),
           "value": b_csr(Nloc, Nloc * (1 + Ns)) / (dt / K) / 1e9, "unit": "GB/s (equivalent-CSR bytes of the rank's rows / step time)",
           "emulated": {"rank": r, "of": P, "exchange": "device-to-device copies of the received byte count on the comm stream"},
           "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "1-orbital Anderson impurity, Nbath=%d (Ns=%d), N=%d, rank %d of %d owns %d dw columns = %d rows, storage=%s" % (a.nbath, Ns, N, r, P, int(G.info.mpiQdw), Nloc, a.storage),
                      "storage": a.storage, "N": N, "rows_of_rank": Nloc, "forms": G.describe()},
           "time_split_ms": {"hxv_product_of_rank": hxv_ms, "axpy_norm_reductions_scalars": dt / K * 1e3 - hxv_ms, "emulated_exchange_on_comm_stream": comm_ms},
           "roofline_of_rank": {"algorithmic_bytes": bytes_fmt, "achieved_GBps": bytes_fmt / (hxv_ms * 1e-3) / 1e9 if hxv_ms > 0 else None, "frac": bytes_fmt / (hxv_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if hxv_ms > 0 else None},
           "exchange": {"mode": xmode, "received_MB_per_product": round(float(recvb.sum()) / 1e6, 2), "sent_MB_per_product": round(float(sendb.sum()) / 1e6, 2),
                        "busiest_peer_received_MB": round(float(recvb.max()) / 1e6, 2), "received_by_peer_MB": [round(float(x) / 1e6, 2) for x in recvb]},
           "model_on_xgmi_ms": {"busiest_peer_at_77GBps": float(recvb.max()) / 77e9 * 1e3, "busiest_peer_at_60GBps": float(recvb.max()) / 60e9 * 1e3,
                                "balanced_over_%d_links_at_77GBps" % max(P - 1, 1): float(recvb.sum()) / max(P - 1, 1) / 77e9 * 1e3,
                                "note": "link times are a model; kernel times above are measured"},
           "build_s": t_build}
    print(json.dumps(out))
    G.close(); comm.close()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nbath", type=int, default=15, help="Ns = nbath+1 (15 -> BASELINE config 3; 17 -> config 4, use --storage direct)")
    ap.add_argument("--storage", default="csr", choices=["csr", "kron", "direct"])
    ap.add_argument("--no-extra", action="store_true", help="skip the kron/direct side measurements")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-ref1", action="store_true", help="N>1: skip the one-GPU reference timing of the same sector on rank 0")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch", "gloo"], help="N>1: communicator owned by libedgpu.so (rccl), torch.distributed device callbacks (torch), or host-staged gloo callbacks (gloo: functional tests, several ranks on one GPU)")
    ap.add_argument("--force-dist", action="store_true", help="run the N>1 code path (process group, communicator, exchange) even with one rank (self-test)")
    ap.add_argument("--share-gpus", action="store_true", help="allow more ranks than devices (ranks share devices; implies --comm gloo)")
    ap.add_argument("--launch-timeout", type=int, default=1500)
    ap.add_argument("--emulate-rank", type=int, default=None, help="with --of P: time ONE rank of a P-rank job alone on this GPU (emulated communicator: exchanges become device copies of the same byte count); a measurement aid, not a job throughput")
    ap.add_argument("--of", type=int, default=8, help="rank count of the emulated job")
    a = ap.parse_args()
    if a.emulate_rank is not None:
        raise SystemExit(emulate_rank(a))
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.share_gpus:
        a.comm = "gloo"
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:      # no launcher: be the launcher
        raise SystemExit(launch_ranks(a, sys.argv[1:]))

    # exactly ONE line on stdout: libraries that chat on fd 1 (RCCL prints a version banner at communicator set-up)
    # are sent to stderr for the duration of the run; the JSON goes to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; refusing to report a different GPU count\n" % (a.gpus, world))
        raise SystemExit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libedgpu.so has no CPU path)")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        if not a.share_gpus:
            raise SystemExit("bench.py: LOCAL_RANK %d but only %d device(s) visible" % (local_rank, ndev))
        local_rank %= ndev
    torch.cuda.set_device(local_rank)
    dist = None
    multi = world > 1 or a.force_dist
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if a.comm == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import numpy as np
    import edgpu
    from edgpu import model

    m, nu, nd = model.half_filled_c(a.nbath)
    t_build = time.perf_counter()
    G = edgpu.HvSector(m, nu, nd, storage=a.storage, rank=rank, size=world, device=local_rank, force_exchange=a.force_dist)
    comm_used = "none"; comm = None; n_ranks_seen = 1
    if multi:
        ok = torch.ones(1, device="cuda")
        if a.comm == "rccl":
            # ONE RCCL communicator per process, owned by libedgpu.so (id broadcast by torch.distributed), attached to the sector
            try:
                uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
                if rank == 0:
                    uid.copy_(torch.frombuffer(bytearray(edgpu.rccl_unique_id()), dtype=torch.uint8))
                dist.broadcast(uid, 0)
                comm = edgpu.Communicator.rccl(bytes(uid.cpu().numpy().tobytes()), rank, world, device=local_rank)
            except Exception as e:  # keep the job alive: every rank must take the same branch
                print("rank %d: native RCCL communicator failed (%s)" % (rank, e), file=sys.stderr)
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if a.comm == "gloo":
            from edgpu import exchange
            ex = exchange.device_staged(dist)
            comm = edgpu.Communicator.callbacks(rank, world, ex.allgatherv, ex.exchange, ex.allreduce)
            comm_used = "host-staged gloo callbacks (functional test path)"
        elif a.comm == "torch" or ok.item() == 0:
            from edgpu import exchange
            ex = exchange.TorchDeviceExchange(dist)
            comm = edgpu.Communicator.callbacks(rank, world, ex.allgatherv, ex.exchange, ex.allreduce)
            comm_used = "torch.distributed(nccl) device callbacks"
        else:
            comm_used = "native RCCL communicator (ncclSend/ncclRecv group + ncclAllReduce)"
        G.attach_comm(comm)
        n_ranks_seen = int(round(float(comm.allreduce_sum(np.ones(1))[0])))     # every rank adds 1 through the communicator the products use
        if n_ranks_seen != world:
            sys.stderr.write("bench.py: the communicator sums %d ranks, WORLD_SIZE is %d\n" % (n_ranks_seen, world))
            raise SystemExit(4)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build

    K, W = a.steps, a.warmup
    dt = timed_steps(G, K, W, torch, dist)
    hxv_ms, alphas, betas = G.bench_result(K, ncoef=W + K)
    comm_ms = G.bench_comm_ms(K) if multi else 0.0
    xmode, sendb, recvb = G.exchange_info()

    N = int(G.Dim); Ns = m["Ns"]
    nnz = N * (1 + Ns)                   # 1 diag + Ns/2 up + Ns/2 dw per row at half filling, all V != 0
    Nloc = int(G.vecDim); nnz_loc = int(G.info.nnz_total)
    ms_step = dt / K * 1e3
    bytes_fmt = {"csr": b_csr(Nloc, nnz_loc), "kron": 24 * Nloc, "direct": 16 * Nloc}[a.storage]
    kernel_key = a.storage
    bytes_bracket = bytes_fmt
    par = "single GPU"
    if multi:
        par = "dw-column split x%d (ED_HAMILTONIAN.f90:96-110), %s exchange per product over %s" % (
            world, {"needed-rows": "needed-rows (send/recv pairs)", "all-gather": "all-gather of the input vector",
                    "transpose": "distributed-transpose (two all-to-alls)", "none": "no"}[xmode], comm_used)
    out = {
        "metric": "Lanczos HxV GB/s (equivalent-CSR bytes of one product / Lanczos-iteration time), Ns=%d half-filled sector" % Ns,
        "value": b_csr(N, nnz) / (dt / K) / 1e9, "unit": "GB/s",
        "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": K, "warmup": W, "ms_per_step": ms_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "iters_per_sec": K / dt,
        "config": {"workload": "1-orbital Anderson impurity, Nbath=%d (Ns=%d), sector Nup=Ndw=%d, N=%d, default initial bath, operator storage=%s, one Lanczos iteration per step"
                               % (a.nbath, Ns, nu[0], N, a.storage),
                   "storage": a.storage, "N": N, "nnz": nnz, "parallelism": par, "forms": G.describe()},
        "roofline": {"bound": "hbm", "kernel": KERNELS[kernel_key],
                     "achieved": bytes_bracket / (hxv_ms * 1e-3) / 1e9 if hxv_ms > 0 else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": (bytes_bracket / (hxv_ms * 1e-3) / 1e9) / HBM_PEAK_GBPS if hxv_ms > 0 else None,
                     "algorithmic_bytes_per_launch": bytes_bracket, "kernel_ms": hxv_ms, "traffic": None,
                     "algorithmic_bytes_are": "algorithmic bytes of one product in the storage that is streamed (SURVEY 8d); for storage=csr: B_csr = 12*nnz + 8*(N+1) + 16*N",
                     "bound_note": BOUND_NOTE.get(a.storage)},
        "time_split_ms": dict({"hxv_product": hxv_ms, "axpy_norm_reductions_scalars": ms_step - hxv_ms},
                              **({"exchange_on_comm_stream_overlapped_with_local_blocks": comm_ms} if multi else {})),
        "build_s": t_build,
        "lanczos_alpha_first_last": [float(alphas[0]), float(alphas[-1])] if len(alphas) else None,
    }
    if multi:      # who sends how much to whom per product (row r = bytes rank r sends to every peer): link balance at a glance
        mat = torch.zeros(world, world, dtype=torch.float64, device="cuda" if a.comm != "gloo" else "cpu")
        mat[rank] = torch.as_tensor(sendb.astype(np.float64), device=mat.device)
        dist.all_reduce(mat)
        out["exchange_bytes_per_product"] = {"mode": xmode, "send_matrix_MB": [[round(float(x) / 1e6, 2) for x in row] for row in mat.cpu().tolist()],
                                             "busiest_pair_MB": round(float(mat.max()) / 1e6, 2), "per_rank_received_MB": [round(float(x) / 1e6, 2) for x in mat.sum(0).cpu().tolist()]}
        # what that exchange should cost on xGMI (point-to-point links, one message per peer and direction; 60-77 GB/s per
        # direction and link measured elsewhere on this generation): the busiest pair bounds the product.
        # A model to read the first real multi-GPU run against -- NOT a measurement.
        bp = float(mat.max())
        out["exchange_model_ms"] = {"busiest_pair_at_77GBps": bp / 77e9 * 1e3, "busiest_pair_at_60GBps": bp / 60e9 * 1e3,
                                    "perfectly_balanced_over_%d_links_at_77GBps" % max(world - 1, 1): float(mat.sum(0).max()) / max(world - 1, 1) / 77e9 * 1e3,
                                    "measured_exchange_ms_on_comm_stream": comm_ms, "note": "model, not a measurement; ranks sharing one GPU exchange through device copies"}
        if xmode == "needed-rows":      # what the opt-in relay routing (EDGPU_RELAY=1, two phases through idle peers) does or would do to the busiest link
            ri = G.relay_info()
            rowb = float(G.DimUp) * 8.0
            out["exchange_model_ms"]["relay_routing"] = {"enabled": ri["enabled"], "relayed_chunks": ri["chunks"],
                                                         "busiest_link_MB": round(ri["busiest_after"] * rowb / 1e6, 2),
                                                         "busiest_link_at_77GBps": ri["busiest_after"] * rowb / 77e9 * 1e3}
    # HBM traffic from the counter passes committed under profiles/: only while it was measured on THIS code
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            stamp = source_stamp()
            for rec in json.load(open(pmc)):
                if rec.get("storage") == a.storage and rec.get("nbath") == a.nbath and rec.get("n_gpus") == world:
                    if rec.get("source_stamp") == stamp:
                        out["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
                        out["roofline"]["traffic_source"] = rec.get("source")
                        out["roofline"]["traffic_kernels"] = rec.get("kernels")
                        if rec.get("rocprof_kernel_ms_per_product"):      # the same kernels under rocprofv3 --kernel-trace (profiles/): a few % slower than by HIP events
                            out["roofline"].setdefault("committed_profile", {})["what"] = "numbers of the committed rocprofv3 round under profiles/ (builder's box, NOT this run): kernel time per step and the fraction it gives"
                            out["roofline"]["committed_profile"]["file"] = rec.get("file")
                            out["roofline"]["committed_profile"]["kernel_ms"] = rec["rocprof_kernel_ms_per_product"]
                            out["roofline"]["committed_profile"]["frac"] = bytes_bracket / (rec["rocprof_kernel_ms_per_product"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
                    else:
                        out["roofline"]["traffic_source"] = "null: the committed counter summary (%s) was measured on source stamp %s, the running code is %s" % (
                            rec.get("file", "profiles/"), rec.get("source_stamp"), stamp)
        except Exception:
            pass
    if world > 1 and not a.no_ref1:
        # the same sector on ONE GPU (rank 0 alone, the others wait): the reference point of the strong-scaling ratio
        ref = None
        if rank == 0:
            try:
                G.close(); G = None
                g1 = edgpu.HvSector(m, nu, nd, storage=a.storage, device=local_rank)
                d1 = timed_steps(g1, K, W, torch, None)
                g1.close()
                ref = {"ms_per_step_1gpu": d1 / K * 1e3, "speedup_vs_1gpu": (d1 / K) / (dt / K)}
                if Ns == 18:
                    ref["ns18_ref_ms_1gpu"] = d1 / K * 1e3
            except Exception as e:
                ref = {"error": "one-GPU reference run failed: %s" % e}
        dist.barrier()
        if rank == 0:
            out["strong_scaling_reference"] = ref
    if world == 1 and rank == 0 and not a.no_extra:
        extra = {}
        # what a real run sees: a ground-state search is >= 89 steps, a Green's-function leg 200 (lanc_niter 512, ED_INPUT_VARS.f90:188):
        # 200 timed steps after 50 warm-up steps (the device slows under sustained load; `value` above stays the contract's K steps)
        try:
            ds = timed_steps(G, 200, 50, torch, None)
            hs, _, _ = G.bench_result(200)
            out["sustained"] = {"steps": 200, "warmup": 50, "ms_per_step": ds / 200 * 1e3, "value": b_csr(N, nnz) / (ds / 200) / 1e9, "unit": "GB/s", "kernel_ms": hs,
                                "frac": (bytes_bracket / (hs * 1e-3) / 1e9) / HBM_PEAK_GBPS if hs > 0 else None}
        except Exception as e:
            out["sustained"] = {"error": str(e)}
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
                    dv.zero()
                    t0 = time.perf_counter(); e0, nl = g.sp_lanc_eigh_device(dv, nitermax=300, threshold=1e-12); tg = time.perf_counter() - t0
                dv.free()
                out["ground_state"] = {"storage": "kron", "E0": e0, "nlanc": nl, "seconds": tg}
                # the reference's DEFAULT solver path: lanc_method=arpack -> sp_eigh(Neigen=2, Nblock=10*max(Neigen,lanc_nstates_sector))
                # (ED_INPUT_VARS.f90:182-187, ED_DIAG.f90:95-98,153-170) = edgpu_eigsh(nev=2, ncv=20); second call (buffers allocated)
                edgpu.cache_release()
                for _ in range(2):
                    t0 = time.perf_counter(); ev, _x, info = g.sp_eigh(2, nblock=20, nitermax=300, tol=1e-10, vectors=False); te = time.perf_counter() - t0
                out["eigsh_c3_s"] = te
                out["default_solver_path"] = {"call": "edgpu_eigsh(nev=2, ncv=20, tol=1e-10) = sp_eigh of ED_DIAG.f90:153-170 (the library may enlarge the basis when HBM is free: ncv_used)", "storage": "kron", "seconds": te,
                                              "E": [float(x) for x in ev], "products": info["matvecs"], "restarts": info["restarts"], "ncv_used": info["ncv_used"]}
                # lanc_method=dvdson (ED_DIAG.f90:189-203) on the same sector: edgpu_dvdson, second call; never allowed to cost the line
                try:
                    for _ in range(2):
                        t0 = time.perf_counter(); evd, _x, infd = g.sp_dvdson_eigh(2, nitermax=600, tol=1e-10, vectors=False); td = time.perf_counter() - t0
                    out["dvdson_c3_s"] = td
                    out["dvdson_path"] = {"call": "edgpu_dvdson(nev=2, tol=1e-10) = sp_dvdson_eigh of ED_DIAG.f90:189-203", "storage": "kron", "seconds": td,
                                          "E": [float(x) for x in evd], "products": infd["matvecs"]}
                except Exception as e:      # noqa: BLE001
                    out["dvdson_c3_s"] = None; out["dvdson_path"] = {"error": str(e)[:200]}
            g.close()
        out["other_formats"] = extra
        # the configurations the driver never runs: BASELINE config 4 (Ns=18, on-the-fly, the whole sector on ONE GPU) and config 5
        # (2-orbital replica bath, Ns=16, stored factors) -- one-line timings next to the headline
        if a.nbath == 15:      # only beside the headline workload (the contract tests run tiny sectors)
            try:
                m4, nu4, nd4 = model.half_filled_c(17)
                g4 = edgpu.HvSector(m4, nu4, nd4, storage="direct", device=local_rank)
                N4 = int(g4.Dim)
                v4 = edgpu.DeviceVector(N4); h4 = edgpu.DeviceVector(N4)
                v4.zero(); h4.zero()
                ms4 = g4.hxv_time(v4, h4, 1, 4)
                v4.free(); h4.free(); g4.close()
                out["c4_direct_product_ms"] = ms4
                out["c4_direct_product"] = {"config": "BASELINE config 4: Ns=18, N=%d, on-the-fly operator, one GPU" % N4, "ms": ms4, "format_GBps": 16 * N4 / (ms4 * 1e-3) / 1e9, "frac_of_peak": 16 * N4 / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBPS}
            except Exception as e:
                out["c4_direct_product"] = {"error": str(e)}
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import cases
                m5 = model.make_model(Norb=2, Nbath=7, bath_type="replica", H_basis=[cases.sigma_z()], lambda_imp=[1.0], Uloc=(2.0, 2.0), Jh=0.25, Ust=1.5, Jx=0.0, Jp=0.0)
                g5 = edgpu.HvSector(m5, [8], [8], storage="kron", device=local_rank)
                g5.bench_begin(16); g5.bench_steps(3); torch.cuda.synchronize()
                t0 = time.perf_counter(); g5.bench_steps(10); torch.cuda.synchronize(); it5 = (time.perf_counter() - t0) / 10
                N5 = int(g5.Dim); g5.close()
                out["c5_kron_iteration_ms"] = it5 * 1e3
                out["c5_kron_iteration"] = {"config": "BASELINE config 5: 2-orbital replica bath, Ns=16, N=%d, stored factors, Jx=Jp=0" % N5, "ms": it5 * 1e3}
            except Exception as e:
                out["c5_kron_iteration"] = {"error": str(e)}
    if world == 1 and rank == 0 and not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if G is not None:
        G.close()
    if comm is not None:
        comm.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
