"""The Fortran test scaffold (tests/fortran_scaffold/) declares the host variables the same-name drop-in modules USE with the
same type, kind, rank and attributes as the reference (scripts/check_scaffold_conformance.py).  Runs where /root/reference
exists (the build container); it reads the reference's declarations as text, nothing is compiled or copied."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def _script():
    spec = importlib.util.spec_from_file_location("check_scaffold_conformance", os.path.join(ROOT, "scripts", "check_scaffold_conformance.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present on this machine")
def test_scaffold_declarations_match_the_reference(monkeypatch, capsys):
    mod = _script()
    monkeypatch.setattr("sys.argv", ["check", "--reference", REF])
    assert mod.main() == 0, capsys.readouterr().out


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present on this machine")
@pytest.mark.parametrize("old,new", [
    ("real(8),dimension(:,:,:,:),allocatable           :: impHloc", "real(8),dimension(:,:,:),allocatable           :: impHloc"),      # rank
    ("logical                                          :: Jhflag", "integer                                          :: Jhflag"),      # type
    ("integer,allocatable,dimension(:)          :: DimUps", "integer,dimension(5)          :: DimUps"),                                # attribute
    ("real(8),dimension(:),allocatable :: lambda", "real(8),dimension(:),allocatable :: lambdas"),                                      # component name
    ("subroutine dd_sparse_HxV(Nloc,v,Hv)", "subroutine dd_sparse_HxV(Nloc,Hv,v)"),                                                    # interface
])
def test_conformance_script_detects_drift(monkeypatch, tmp_path, capsys, old, new):
    mod = _script()
    src = open(mod.SCAFFOLD).read()
    assert old in src, old
    p = tmp_path / "scaffold_modules.f90"
    p.write_text(src.replace(old, new))
    monkeypatch.setattr(mod, "SCAFFOLD", str(p))
    monkeypatch.setattr("sys.argv", ["check", "--reference", REF])
    assert mod.main() == 1
    assert "MISMATCH" in capsys.readouterr().out
