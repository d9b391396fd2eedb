"""The stopping rule of sp_lanc_eigh (ED_DIAG.f90:177-185 -> SciFortran's sp_lanc_eigh: lowest eigenvalue of T_k after every iteration,
stop when it moved by less than the threshold) as the device loops apply it: a pivot-based estimate of the lowest eigenvalue first, the QL
sweep only when the estimate is within a wide margin of the threshold.  Host arithmetic only (csrc/tridiag_host.hpp compiled with g++):
on random tridiagonals and on tridiagonals produced by a Lanczos run the rule fires at the same iteration as with a QL sweep per
iteration, for thresholds 1e-10 ... 1e-14, and the estimate agrees with QL to a few 1e-15."""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stopping_rule_estimate_never_changes_the_iteration(tmp_path):
    exe = str(tmp_path / "stoprule_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "stoprule_check.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "mismatches 0" in r.stdout and "Lanczos-generated: 0 mismatches of 240" in r.stdout
    rel = float(r.stdout.split("max rel diff estimate vs QL")[1].split()[0])
    assert rel < 5e-14
