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
