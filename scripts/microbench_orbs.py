"""ed_total_ud=F product timing (development aid): python scripts/microbench_orbs.py <Nbath per orbital> <storage> [nrep]
Norb=2 normal bath, per-orbital half filling; EDGPU_COMPOSITE=0 runs the generic mixed-radix kernel instead of the composite two-factor path."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model
nb = int(sys.argv[1]); st = sys.argv[2]; nrep = int(sys.argv[3]) if len(sys.argv) > 3 else 10
m = model.make_model(Norb=2, Nbath=nb, ed_total_ud=False, Uloc=(2.0, 2.0), Ust=1.0, Jh=0.2)
n = (nb + 1) // 2
G = edgpu.HvSector(m, [n, n], [n, n], storage=st)
N = G.Dim
v = edgpu.DeviceVector(N); h = edgpu.DeviceVector(N)
v.upload(np.random.RandomState(0).uniform(-0.5, 0.5, N))
ms = G.hxv_time(v, h, 2, nrep)
nnz = G.info.nnz_total
bcsr = 12 * nnz + 8 * (N + 1) + 16 * N
knobs = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("EDGPU_"))
print("orbs Norb=2 nbath=%d N=%d st=%s %s : %.4f ms  eqcsr %.0f GB/s (B_csr %.3f GB)" % (nb, N, st, knobs, ms, bcsr / ms / 1e6, bcsr / 1e9))
