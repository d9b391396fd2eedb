"""edgpu -- host-side mirror (Python) of the MI355X Lanczos-ED hot path.

The compute lives in libedgpu.so (dmft-lanc-ed_amd/csrc, HIP for gfx950) behind the C ABI
declared in include/edgpu.h.  Importing this package does not load the library; the first
call that needs it does, and fails loudly if it has not been built.
"""
from . import model  # noqa: F401
from .hamiltonian import HvSector, DeviceVector, Communicator, comm_count, build_Hv_sector, device_count, rccl_unique_id, apply_op, gf_lanczos, cache_release  # noqa: F401
