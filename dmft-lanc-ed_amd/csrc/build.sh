#!/bin/bash
# Build libedgpu.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
#   build.sh        the product
#   build.sh exp    additionally libedgpu_exp.so with -DEDGPU_EXPERIMENTS: tuning values become environment switches and the
#                   timing-only debug bits exist (scripts/ select it with EDGPU_LIBRARY=.../libedgpu_exp.so); never loaded by tests or bench.py
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o "$HERE/../edgpu/libedgpu.so" "$HERE/edgpu.hip" -ldl
if [ "$1" = "exp" ]; then
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DEDGPU_EXPERIMENTS -o "$HERE/../edgpu/libedgpu_exp.so" "$HERE/edgpu.hip" -ldl
fi
