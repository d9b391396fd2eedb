import sys
sys.path[:0]=['/root/repo','/root/repo/dmft-lanc-ed_amd']
import numpy as np, edgpu
from edgpu import model
m = model.make_model(Norb=1, Nbath=1, Uloc=(0.0,), bath_e=np.zeros((1, 1, 1)), bath_v=np.full((1, 1, 1), 0.5))
G = edgpu.HvSector(m,[1],[1])
print(np.linalg.eigvalsh(G.hmat()))
print(G.sp_eigh(2, nblock=4, tol=1e-13, allow_unconverged=True))
