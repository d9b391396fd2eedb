import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))
import numpy as np
import torch
import edgpu
v = edgpu.DeviceVector(1000).upload(np.arange(1000, dtype=np.float64))
class W:
    def __init__(self, ptr, n): self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}
t = torch.as_tensor(W(v.ptr, 1000), device="cuda")
print("wrapped sum", float(t.sum()), "expected", 999*1000/2)
t.mul_(2.0); torch.cuda.synchronize()
print("after in-place mul via torch, edgpu download:", v.download()[:4])
with open("/proc/self/maps") as f:
    libs = sorted(set(l.split()[-1] for l in f if "libamdhip64" in l or "librccl" in l))
print(libs)
