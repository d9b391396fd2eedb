#!/bin/bash
# Compile the ISO_C_BINDING shim modules and the config-1 demo driver with amdflang (the
# reference's host language) against the in-tree libedgpu.so.  Skipped when amdflang is absent.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
FC="$(command -v amdflang || command -v /opt/rocm/bin/amdflang || true)"
if [ -z "$FC" ]; then echo "amdflang not found: Fortran shim not built"; exit 0; fi
mkdir -p "$HERE/_build"
cd "$HERE/_build"
"$FC" -O2 -c "$HERE/edgpu_c_api.f90" -o edgpu_c_api.o
"$FC" -O2 -c "$HERE/ED_HAMILTONIAN_GPU.f90" -o ED_HAMILTONIAN_GPU.o
"$FC" -O2 -c "$HERE/ED_SPARSE_MATRIX_GPU.f90" -o ED_SPARSE_MATRIX_GPU.o
"$FC" -O2 "$HERE/demo_c1.f90" edgpu_c_api.o ED_HAMILTONIAN_GPU.o ED_SPARSE_MATRIX_GPU.o -L"$HERE/../edgpu" -ledgpu -Wl,-rpath,'$ORIGIN/../../edgpu' -o demo_c1
# same-name drop-in of the reference module ED_SPARSE_MATRIX and a host-only demo of it
"$FC" -O2 -c "$HERE/dropin/ED_SPARSE_MATRIX.f90" -o ED_SPARSE_MATRIX.o
"$FC" -O2 "$HERE/dropin/demo_sparse_dropin.f90" edgpu_c_api.o ED_SPARSE_MATRIX_GPU.o ED_SPARSE_MATRIX.o -L"$HERE/../edgpu" -ledgpu -Wl,-rpath,'$ORIGIN/../../edgpu' -o demo_sparse_dropin
echo "built $HERE/_build/demo_c1"
