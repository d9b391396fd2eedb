#!/usr/bin/env python3
"""hipMalloc / first-touch cost on the box (decides how big work buffers are handled): alloc_time.py"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
from edgpu import capi
if len(sys.argv) > 1 and sys.argv[1] == "torch": import torch
L = capi.lib()
hip = C.CDLL("libamdhip64.so")
def malloc(gb):
    p = C.c_void_p(); t0 = time.perf_counter(); capi.check(L.edgpu_malloc(C.byref(p), C.c_size_t(int(gb * 1e9)))); return p, time.perf_counter() - t0
def touch(p, gb):
    t0 = time.perf_counter(); hip.hipMemset(p, 0, C.c_size_t(int(gb * 1e9))); hip.hipDeviceSynchronize(); return time.perf_counter() - t0
for gb in (1.3, 21.0):
    p, ta = malloc(gb); t1 = touch(p, gb); t2 = touch(p, gb)
    L.edgpu_free(p)
    q, tb = malloc(gb); t3 = touch(q, gb); t4 = touch(q, gb)
    L.edgpu_free(q)
    print("%5.1f GB fresh: malloc %.1f ms, 1st memset %.1f ms, 2nd %.1f ms | after free: malloc %.1f ms, 1st memset %.1f ms, 2nd %.1f ms" % (gb, ta * 1e3, t1 * 1e3, t2 * 1e3, tb * 1e3, t3 * 1e3, t4 * 1e3), flush=True)
