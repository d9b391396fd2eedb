import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dmft-lanc-ed_amd"))
import edgpu
from edgpu import model
# warm-up: HIP initialisation and kernel loading are not part of a sector's create time (round-3 review: they were charged to the first row)
_m, _nu, _nd = model.half_filled_c(5); _g = edgpu.HvSector(_m, _nu, _nd, storage="kron"); _g.spHtimesV(__import__("numpy").ones(_g.Dim)); _g.close()
for nb, sts in ((15, ("kron", "direct", "csr")), (17, ("kron", "direct"))):
    m, nu, nd = model.half_filled_c(nb)
    for st in sts:
        t0 = time.perf_counter(); G = edgpu.HvSector(m, nu, nd, storage=st); t1 = time.perf_counter() - t0
        t0 = time.perf_counter(); G.close(); t2 = time.perf_counter() - t0
        print("Ns=%d %s: create %.3f s, destroy %.3f s" % (nb + 1, st, t1, t2), flush=True)
