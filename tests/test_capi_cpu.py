"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/edgpu.h
declares, and refuses to compute without a GPU (no silent fallback)."""
import ctypes as C
import os
import re

import pytest

from edgpu import capi, model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "edgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(edgpu_[a-z0-9_]+)\s*\(", src)) - {"edgpu_allgatherv_fn", "edgpu_allreduce_fn"})


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    declared = _header_symbols()
    assert declared, "no declarations parsed"
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(capi.SYMBOLS) == declared
    assert L.edgpu_abi_version() == capi.EDGPU_ABI_VERSION


def test_struct_layouts_match_header():
    # sizes the C compiler gives the two structs (guards against ctypes/ABI drift)
    import subprocess, tempfile
    code = '#include "edgpu.h"\n#include <stdio.h>\nint main(){printf("%zu %zu\\n",sizeof(edgpu_desc),sizeof(edgpu_info));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c"); exe = os.path.join(d, "s")
        open(src, "w").write(code)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        a, b = map(int, subprocess.check_output([exe]).split())
    assert C.sizeof(capi.Desc) == a and C.sizeof(capi.Info) == b


def test_compute_fails_loudly_without_gpu():
    import edgpu
    if edgpu.device_count() > 0:
        pytest.skip("a GPU is visible")
    m, nu, nd = model.half_filled_c(5)
    with pytest.raises(capi.EdgpuError) as ei:
        edgpu.HvSector(m, nu, nd)
    assert ei.value.code == capi.ENODEVICE
    assert "no CPU fallback" in str(ei.value)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "dmft-lanc-ed_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".f90")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), os.path.join(dp, f)


def test_dw_partition_matches_reference_formula():
    """ED_HAMILTONIAN.f90:96-110: first mod(DimDw,P) ranks own one more column."""
    for DimDw, P in [(12870, 8), (20, 3), (7, 7), (924, 5)]:
        tot = 0
        for r in range(P):
            q, d0 = model.dw_partition(DimDw, P, r)
            mpiQdw = DimDw // P + (1 if r < DimDw % P else 0)
            mpiRdw = 0 if r < DimDw % P else DimDw % P
            assert q == mpiQdw and d0 == r * mpiQdw + mpiRdw
            tot += q
        assert tot == DimDw


_ORDER_SNIPPET = """
import sys, ctypes as C
sys.path.insert(0, %(pkg)r)
from edgpu import capi
%(first)s
%(second)s
rt = capi.mapped_hip_runtimes()
p = C.c_void_p()
rc = capi.lib().edgpu_malloc(C.byref(p), 8)          # first entry that would touch the device
print("RUNTIMES", len(rt)); print("RC", rc); print("MSG", capi.lib().edgpu_last_error().decode())
%(tail)s
"""


def _run_order(first, second, env=None, tail=""):
    import subprocess, sys
    code = _ORDER_SNIPPET % dict(pkg=os.path.join(ROOT, "dmft-lanc-ed_amd"), first=first, second=second, tail=tail)
    e = dict(os.environ); e.pop("EDGPU_HIP_RUNTIME", None); e.update(env or {})
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=e, timeout=600)
    return r.returncode, r.stdout.decode()


def test_one_hip_runtime_whatever_the_import_order():
    """Round-2 abort ("double free or corruption" at interpreter exit) came from two HIP runtimes in one process: libedgpu.so
    bound to /opt/rocm's libamdhip64, then `import torch` mapping PyTorch's own copy.  The binding now loads the embedding
    framework's runtime first, so both orders end with ONE runtime and a clean exit."""
    pytest.importorskip("torch")
    for first, second in (("capi.lib()", "import torch"), ("import torch", "capi.lib()")):
        rc, out = _run_order(first, second)
        assert rc == 0, out
        assert "RUNTIMES 1" in out, out
        assert "copies of the HIP runtime" not in out, out


def test_two_hip_runtimes_are_refused_with_a_clear_error():
    """EDGPU_HIP_RUNTIME=system keeps the ROCm installation's runtime; a PyTorch imported afterwards brings a second one.
    The library then refuses every entry that would touch the device (EDGPU_EHIP and the two paths) instead of running into
    the exit-time heap corruption, and the Python binding raises the same complaint when it is asked to check."""
    pytest.importorskip("torch")
    tail = "\ntry:\n    capi.assert_single_hip_runtime()\nexcept ImportError as e:\n    print('PYERR', e)\n"
    rc, out = _run_order("capi.lib()", "import torch", env={"EDGPU_HIP_RUNTIME": "system"}, tail=tail)
    assert rc == 0, out
    if "RUNTIMES 1" in out:
        pytest.skip("this PyTorch shares the system HIP runtime")
    assert "RC -3" in out and "copies of the HIP runtime are mapped" in out and "PYERR" in out, out
