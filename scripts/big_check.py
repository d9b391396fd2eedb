"""Large-sector checks on one GPU (development aid): python scripts/big_check.py <nbath> <storage> [lanczos_iters]
Size-independent properties only (no oracle at these sizes): symmetry <u,Hv>=<Hu,v>, and a Lanczos run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import edgpu
from edgpu import model
nbath = int(sys.argv[1]); st = sys.argv[2]; nit = int(sys.argv[3]) if len(sys.argv) > 3 else 0
m, nu, nd = model.half_filled_c(nbath)
t0 = time.time(); G = edgpu.HvSector(m, nu, nd, storage=st); N = G.Dim
print("N=%d  create %.2fs  bytes_stored=%.2f GB" % (N, time.time() - t0, G.info.bytes_stored / 1e9), flush=True)
rng = np.random.default_rng(1)
u = rng.random(N) - 0.5; v = rng.random(N) - 0.5
du, dv, dh = edgpu.DeviceVector(N).upload(u), edgpu.DeviceVector(N).upload(v), edgpu.DeviceVector(N)
G.spHtimesV_device(dv, dh); hv = dh.download()
G.spHtimesV_device(du, dh); hu = dh.download()
a, b = float(u @ hv), float(hu @ v)
print("symmetry: <u,Hv>=%.12e <Hu,v>=%.12e rel diff %.2e" % (a, b, abs(a - b) / (np.abs(u) @ np.abs(hv))), flush=True)
ms = G.hxv_time(dv, dh, 2, 5)
print("HxV %.3f ms  (16N: %.0f GB/s, 24N: %.0f GB/s)" % (ms, 16 * N / ms / 1e6, 24 * N / ms / 1e6), flush=True)
if nit:
    t0 = time.time(); e0, nl = G.sp_lanc_eigh_device(dv.upload(np.zeros(N)), nitermax=nit, threshold=1e-12)
    x = dv.download(); G.spHtimesV_device(dv, dh); r = dh.download() - e0 * x
    print("lanczos: E0=%.12f nlanc=%d  |x|=%.12f  residual=%.3e  (%.1fs)" % (e0, nl, np.linalg.norm(x), np.linalg.norm(r), time.time() - t0), flush=True)
