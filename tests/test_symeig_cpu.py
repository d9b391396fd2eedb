"""The dense symmetric eigenproblem of the projected matrices (one per restart of edgpu_eigsh, one per step of edgpu_dvdson): Householder
reduction + QL (csrc/tridiag_host.hpp: sym_eig_householder) against cyclic Jacobi sweeps, host arithmetic only (g++): 400 matrices of the
shapes the solvers produce -- dense, Ritz block + arrow + tridiagonal tail (also with couplings of 1e-11), degenerate diagonals,
decoupled tridiagonals, 1 to 96 rows -- eigenvalues to 1e-14 |A|, residuals to 1e-14 |A|, orthonormal vectors, ascending order."""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_householder_ql_matches_jacobi(tmp_path):
    exe = str(tmp_path / "symeig_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "symeig_check.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "cases 400 failures 0" in r.stdout
