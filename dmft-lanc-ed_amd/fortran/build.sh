#!/bin/bash
# Compile the ISO_C_BINDING shim modules, the same-name drop-in modules and the demo drivers with amdflang (the
# reference's host language) against the in-tree libedgpu.so.  Skipped when amdflang is absent.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
FC="$(command -v amdflang || command -v /opt/rocm/bin/amdflang || true)"
if [ -z "$FC" ]; then echo "amdflang not found: Fortran shim not built"; exit 0; fi
mkdir -p "$HERE/_build"
cd "$HERE/_build"
LINK=(-L"$HERE/../edgpu" -ledgpu -Wl,-rpath,'$ORIGIN/../../edgpu')
"$FC" -O2 -c "$HERE/edgpu_c_api.f90" -o edgpu_c_api.o
"$FC" -O2 -c "$HERE/dropin/EDGPU_LIVE.f90" -o EDGPU_LIVE.o
# same-name replacement of the reference module ED_SPARSE_MATRIX (one layer over the C ABI) + a host-only demo of it
"$FC" -O2 -cpp -c "$HERE/dropin/ED_SPARSE_MATRIX.f90" -o ED_SPARSE_MATRIX.o
"$FC" -O2 "$HERE/dropin/demo_sparse_dropin.f90" edgpu_c_api.o EDGPU_LIVE.o ED_SPARSE_MATRIX.o "${LINK[@]}" -o demo_sparse_dropin
# explicit-argument operator module + the config-1 demo driver
"$FC" -O2 -c "$HERE/ED_HAMILTONIAN_GPU.f90" -o ED_HAMILTONIAN_GPU.o
"$FC" -O2 "$HERE/demo_c1.f90" edgpu_c_api.o EDGPU_LIVE.o ED_SPARSE_MATRIX.o ED_HAMILTONIAN_GPU.o "${LINK[@]}" -o demo_c1
# same-name operator modules (ED_HAMILTONIAN_SPARSE_HxV, ED_HAMILTONIAN_DIRECT_HxV) and the SF_SP_LINALG stand-in, compile-
# checked and run against test-only scaffold modules that carry the reference's module names and variables
if [ -d "$ROOT/tests/fortran_scaffold" ]; then
  # an MPI build of the host (-D_MPI): the scaffold supplies `MODULE MPI` with MPI_COMM_NULL, the drop-ins are compiled
  # with the host's preprocessor flags as INTEGRATION.md asks
  MPIF=(-O2 -cpp -D_MPI)
  "$FC" "${MPIF[@]}" -c "$ROOT/tests/fortran_scaffold/scaffold_mpi.f90" -o scaffold_mpi.o
  "$FC" "${MPIF[@]}" -c "$HERE/dropin/ED_SPARSE_MATRIX.f90" -o ED_SPARSE_MATRIX_mpi.o      # replaces the serial .mod from here on
  "$FC" "${MPIF[@]}" -c "$ROOT/tests/fortran_scaffold/scaffold_modules.f90" -o scaffold_modules.o
  "$FC" "${MPIF[@]}" -c "$HERE/dropin/ED_HAMILTONIAN_GPU_BUILD.f90" -o ED_HAMILTONIAN_GPU_BUILD.o
  "$FC" "${MPIF[@]}" -c "$HERE/dropin/ED_HAMILTONIAN_SPARSE_HxV.f90" -o ED_HAMILTONIAN_SPARSE_HxV.o
  "$FC" "${MPIF[@]}" -c "$HERE/dropin/ED_HAMILTONIAN_DIRECT_HxV.f90" -o ED_HAMILTONIAN_DIRECT_HxV.o
  "$FC" "${MPIF[@]}" -c "$HERE/dropin/SF_SP_LINALG.f90" -o SF_SP_LINALG.o
  "$FC" "${MPIF[@]}" "$ROOT/tests/fortran_scaffold/host_like_driver.f90" edgpu_c_api.o EDGPU_LIVE.o scaffold_mpi.o ED_SPARSE_MATRIX_mpi.o scaffold_modules.o \
        ED_HAMILTONIAN_GPU_BUILD.o ED_HAMILTONIAN_SPARSE_HxV.o ED_HAMILTONIAN_DIRECT_HxV.o SF_SP_LINALG.o "${LINK[@]}" -o host_like_driver
fi
echo "built $HERE/_build/demo_c1"
