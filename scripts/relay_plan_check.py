#!/usr/bin/env python3
"""The relay plan (EDGPU_RELAY=1, relay_plan in sector_dev.hpp) on the real row matrices: rank 0 of P ranks of the half-filled
Ns-site sector, on-the-fly storage, no communicator attached (nothing is exchanged).  relay_plan_check.py NBATH P"""
import os, sys
os.environ["EDGPU_RELAY"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dmft-lanc-ed_amd"))
import numpy as np
from edgpu import model
import edgpu
nb, P = int(sys.argv[1]), int(sys.argv[2])
m, nu, nd = model.half_filled_c(nb)
G = edgpu.HvSector(m, nu, nd, storage="direct", rank=0, size=P)
ri = G.relay_info()
D, R = ri["rows_direct"], ri["rows_routed"]
print("Ns=%d P=%d rows per rank %d, received per rank %s" % (nb + 1, P, G.DimDw // P, D.sum(0).tolist()))
print("busiest link %d rows -> %d (two phases), %d relayed chunks; perfectly balanced %.0f" % (ri["busiest_before"], ri["busiest_after"], ri["chunks"], D.sum(0).max() / (P - 1)))
print("rows per link, direct:\n", D, "\nunder the plan (both phases):\n", R)
G.close()
