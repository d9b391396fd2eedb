#!/usr/bin/env python3
"""Is the spread of the explicit-CSR product time (7.85-8.46 ms for the same kernels on the same box) a property of the box or of
where a process's allocations land?  Creates the c3 explicit-CSR sector several times in ONE process (fresh allocations every time)
and times 20 Lanczos steps each; run it twice in a row for the process-to-process spread.  usage: alloc_variance.py [repeats=5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import torch
import edgpu
from edgpu import model
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 5
m, nu, nd = model.half_filled_c(15)
for i in range(rep):
    G = edgpu.HvSector(m, nu, nd, storage="csr")
    G.bench_begin(30); G.bench_steps(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); G.bench_steps(20, timed=True); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    ms, _, _ = G.bench_result(20)
    print("creation %d: product %.3f ms  step %.3f ms" % (i, ms, dt * 1e3), flush=True)
    G.close()
    if i == 1:
        edgpu.cache_release()      # hand the pooled blocks back: the next creation starts from the driver again
