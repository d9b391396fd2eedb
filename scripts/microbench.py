"""Ad-hoc kernel timing sweep (development aid): python scripts/microbench.py <nbath> <storage> [nrep] [lanczos]
(with a 4th argument `lanczos` the timed unit is one Lanczos iteration instead of one product)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model
nbath = int(sys.argv[1]); st = sys.argv[2]; nrep = int(sys.argv[3]) if len(sys.argv) > 3 else 10
m, nu, nd = model.half_filled_c(nbath)
G = edgpu.HvSector(m, nu, nd, storage=st)
N = G.Dim
v = edgpu.DeviceVector(N); h = edgpu.DeviceVector(N)
v.upload(np.random.RandomState(0).uniform(-0.5, 0.5, N))
ms = G.lanczos_time(2, nrep) if len(sys.argv) > 4 and sys.argv[4] == "lanczos" else G.hxv_time(v, h, 2, nrep)
nnz = N * (1 + m["Ns"])
bcsr = 12 * nnz + 8 * (N + 1) + 16 * N
fmt = {"csr": bcsr, "kron": 24 * N, "direct": 16 * N}[st]
knobs = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("EDGPU_"))
print("nbath=%d st=%s %s : %.4f ms  fmt %.0f GB/s  eqcsr %.0f GB/s" % (nbath, st, knobs, ms, fmt / ms / 1e6, bcsr / ms / 1e6))
