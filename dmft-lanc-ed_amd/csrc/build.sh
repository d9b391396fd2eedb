#!/bin/bash
# Build libedgpu.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o "$HERE/../edgpu/libedgpu.so" "$HERE/edgpu.hip" -ldl
