"""The environment switches of the shipping library are a closed, documented set: the table behind edgpu_knob_info must
agree with the call sites in csrc/, with scripts/README.md, with the tests it names and with the strings of the binary (no
name of a removed ablation or tuning switch may survive in libedgpu.so)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dmft-lanc-ed_amd", "csrc")
STRING_NAMES = {"EDGPU_RCCL_LIB", "EDGPU_HIP_RUNTIME", "EDGPU_LIBRARY"}      # path-valued
ERROR_CODES = {"EDGPU_OK", "EDGPU_EINVAL", "EDGPU_ENODEVICE", "EDGPU_EHIP", "EDGPU_ENOMEM", "EDGPU_EUNSUPPORTED", "EDGPU_ECOMM", "EDGPU_ENOTCONVERGED"}


def _table():
    from edgpu import capi
    return capi.knobs()


def _sources():
    out = {}
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hpp", ".hip")):
            out[f] = open(os.path.join(CSRC, f)).read()
    return out


def test_table_equals_call_sites():
    names = {k["name"] for k in _table()}
    used = set()
    for f, src in _sources().items():
        body = src
        if f == "runtime.hpp":      # the table itself and knob()'s own diagnostics
            body = src[src.index("static void warn_unknown_knobs"):]
            used.add("EDGPU_DEBUG")      # dbg_flags()
        used |= set(re.findall(r'\bknob\("(EDGPU_[A-Z0-9_]+)"\)', body))
        assert not re.findall(r'getenv\("EDGPU_(?!RCCL_LIB)', body.replace('getenv(name)', '')), f
    assert used == names, (sorted(used - names), sorted(names - used))
    assert len(names) == len(_table())      # no duplicates


def test_every_switch_names_an_existing_test():
    for k in _table():
        t = k["covered_by"]
        fname, _, func = t.partition("::")
        path = os.path.join(ROOT, "tests", fname)
        assert os.path.exists(path), k
        src = open(path).read()
        assert func and ("def %s(" % func) in src, k
        if k["kind"] != "debug":
            assert k["name"] in src, "%s is not set by %s" % (k["name"], t)


def test_readme_lists_every_switch_with_its_default():
    readme = open(os.path.join(ROOT, "scripts", "README.md")).read()
    for k in _table():
        assert re.search(r"`%s` \(%s\)" % (k["name"], "auto" if k["default"] == -1 else k["default"]), readme), k["name"]


def test_binary_holds_no_other_switch_names():
    lib = os.path.join(ROOT, "dmft-lanc-ed_amd", "edgpu", "libedgpu.so")
    out = subprocess.run(["strings", lib], capture_output=True, text=True, check=True).stdout
    found = set(re.findall(r"EDGPU_[A-Z][A-Z0-9_]*", out))
    allowed = {k["name"] for k in _table()} | STRING_NAMES | ERROR_CODES
    assert found <= allowed, sorted(found - allowed)
    assert "EDGPU_ABL" not in out and "EDGPU_DW_FIRST" not in out


def test_unknown_name_is_reported(tmp_path):
    """warn_unknown_knobs runs at sector creation, i.e. needs a device; without one the entry fails earlier (ENODEVICE).  The
    text of the warning must at least be in the binary."""
    lib = os.path.join(ROOT, "dmft-lanc-ed_amd", "edgpu", "libedgpu.so")
    out = subprocess.run(["strings", lib], capture_output=True, text=True, check=True).stdout
    assert "is not a switch of this library" in out
