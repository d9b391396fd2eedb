"""ctypes binding of libedgpu.so (include/edgpu.h).  No torch types cross this boundary."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EDGPU_LIBRARY: another build of the library (scripts/: the experiments build libedgpu_exp.so); default: the in-tree product
LIB_PATH = os.environ.get("EDGPU_LIBRARY") or os.path.join(_HERE, "libedgpu.so")

EDGPU_ABI_VERSION = 1
MAX_ORB = 5
STORE_KRON, STORE_CSR, STORE_DIRECT = 0, 1, 2
STORAGE = {"kron": STORE_KRON, "csr": STORE_CSR, "direct": STORE_DIRECT}
BATH = {"normal": 0, "hybrid": 1, "replica": 2}
SP_ND, SP_UP, SP_DW = 1, 2, 3
ENODEVICE = -2
ENOMEM = -4
ENOTCONVERGED = -7

# every symbol include/edgpu.h declares
SYMBOLS = [
    "edgpu_sector_create", "edgpu_sector_destroy", "edgpu_sector_vecdim", "edgpu_sector_info",
    "edgpu_sector_map", "edgpu_factor_size", "edgpu_factor_export", "edgpu_diag_export", "edgpu_sector_hmat", "edgpu_csr_rows",
    "edgpu_hxv", "edgpu_hxv64", "edgpu_hxv_device", "edgpu_hxv_time",
    "edgpu_lanczos_gs", "edgpu_lanczos_gs_device", "edgpu_lanczos_tridiag", "edgpu_lanczos_tridiag_device",
    "edgpu_lanczos_time", "edgpu_lanczos_bench_begin", "edgpu_lanczos_bench_steps", "edgpu_lanczos_bench_result", "edgpu_lanczos_bench_comm_ms",
    "edgpu_eigsh", "edgpu_eigsh_ncv_cap", "edgpu_eigsh_ncv_used", "edgpu_lanczos_is_fused", "edgpu_sector_describe", "edgpu_dvdson", "edgpu_cache_release", "edgpu_twin_reorder", "edgpu_imp_density_offdiag", "edgpu_imp_histogram", "edgpu_local_energy", "edgpu_apply_op", "edgpu_gf_lanczos", "edgpu_chi_lanczos", "edgpu_phonon_lanczos",
    "edgpu_sp_init_matrix", "edgpu_sp_init_matrix_mpi", "edgpu_sp_load_matrix", "edgpu_sp_insert_element", "edgpu_sp_delete_matrix", "edgpu_sp_dump_matrix", "edgpu_sp_row", "edgpu_sp_matvec", "edgpu_sp_matvec_device",
    "edgpu_comm_unique_id", "edgpu_comm_init_rccl", "edgpu_comm_init_callbacks",
    "edgpu_comm_create_rccl", "edgpu_comm_create_callbacks", "edgpu_comm_create_emulated", "edgpu_comm_sub", "edgpu_comm_destroy", "edgpu_sector_attach_comm",
    "edgpu_comm_count", "edgpu_comm_allreduce_sum", "edgpu_sector_exchange_info", "edgpu_sector_relay_info",
    "edgpu_device_count", "edgpu_malloc", "edgpu_free", "edgpu_memcpy_h2d", "edgpu_memcpy_d2h", "edgpu_memset",
    "edgpu_device_synchronize", "edgpu_last_error", "edgpu_abi_version", "edgpu_knob_count", "edgpu_knob_info",
]


class Desc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("Norb", C.c_int32), ("Nspin", C.c_int32), ("Nbath", C.c_int32),
        ("Nph", C.c_int32), ("bath_type", C.c_int32), ("ed_total_ud", C.c_int32), ("storage", C.c_int32),
        ("hfmode", C.c_int32), ("Nfoo", C.c_int32), ("nups", C.c_int32 * MAX_ORB), ("ndws", C.c_int32 * MAX_ORB),
        ("mpi_rank", C.c_int32), ("mpi_size", C.c_int32), ("device", C.c_int32), ("reserved", C.c_int32),
        ("Uloc", C.c_double * MAX_ORB), ("Ust", C.c_double), ("Jh", C.c_double), ("Jx", C.c_double),
        ("Jp", C.c_double), ("xmu", C.c_double), ("w0_ph", C.c_double), ("g_ph", C.c_double * MAX_ORB),
        ("impHloc", C.c_void_p), ("diag_hybr", C.c_void_p), ("bath_diag", C.c_void_p), ("hbath", C.c_void_p),
    ]


class Info(C.Structure):
    _fields_ = [
        ("isector", C.c_int32), ("Ns", C.c_int32), ("Ns_Orb", C.c_int32), ("Ns_Ud", C.c_int32), ("DimPh", C.c_int32),
        ("Jhflag", C.c_int32), ("DimUps", C.c_int32 * MAX_ORB), ("DimDws", C.c_int32 * MAX_ORB),
        ("DimUp", C.c_int64), ("DimDw", C.c_int64), ("Dim", C.c_int64), ("mpiQdw", C.c_int64),
        ("mpiIshift", C.c_int64), ("vecDim", C.c_int64), ("nnz_total", C.c_int64), ("bytes_stored", C.c_int64),
    ]


ALLGATHERV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32)

_lib = None
_hip_keep = None


class EdgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("edgpu error %d: %s" % (code, msg))
        self.code = code


def mapped_hip_runtimes():
    """distinct libamdhip64 FILES mapped into this process (/proc/self/maps, one entry per device:inode -- the same library seen
    through two mounts or marked '(deleted)' is one runtime); returns their paths"""
    seen, paths = set(), []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" not in line:
                    continue
                parts = line.split(None, 5)
                if len(parts) < 6:
                    continue
                key = (parts[3], parts[4])
                if key not in seen:
                    seen.add(key); paths.append(parts[5].strip())
    except OSError:
        pass
    return paths


def _preload_embedding_hip_runtime():
    """One HIP runtime per process.  libedgpu.so is linked against libamdhip64.so.7 and the dynamic loader binds it to the
    copy the process already holds -- PyTorch's own when `import torch` came first.  In the other order libedgpu.so would
    bring in /opt/rocm's runtime and a later `import torch` would map PyTorch's copy NEXT to it: two runtimes, correct
    results, and "double free or corruption" at interpreter exit.  So when no runtime is mapped yet and a PyTorch that ships
    one is installed, that copy is loaded first (by path, nothing of torch is imported): whatever the application imports
    afterwards, in whatever order, finds it.  EDGPU_HIP_RUNTIME=system keeps the ROCm installation's runtime (for processes
    that never load PyTorch); the library itself refuses to touch the device while two copies are mapped."""
    if mapped_hip_runtimes() or os.environ.get("EDGPU_HIP_RUNTIME", "auto") == "system":
        return None
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def assert_single_hip_runtime():
    rt = mapped_hip_runtimes()
    if len(rt) > 1:
        raise ImportError("%d copies of the HIP runtime are mapped in this process (%s): import torch BEFORE edgpu; or, in a process that "
                          "imports edgpu first, leave EDGPU_HIP_RUNTIME unset so that edgpu binds to the installed PyTorch's copy; or, when the "
                          "preloaded copy is not the one libedgpu.so resolves to (different SONAME), set EDGPU_HIP_RUNTIME=system and do not "
                          "import torch in that process; continuing would corrupt the heap at exit" % (len(rt), ", ".join(rt)))


def lib():
    """Load libedgpu.so; fails loudly when the HIP extension has not been built."""
    global _lib, _hip_keep
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libedgpu.so is missing (%s): build it with __graft_entry__.build(); there is no fallback path" % LIB_PATH)
    _hip_keep = _preload_embedding_hip_runtime()
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    assert_single_hip_runtime()
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    P = C.POINTER
    sig = {
        "edgpu_sector_create": (C.c_int, [P(Desc), P(vp)]),
        "edgpu_sector_destroy": (C.c_int, [vp]),
        "edgpu_sector_vecdim": (i64, [vp]),
        "edgpu_sector_info": (C.c_int, [vp, P(Info)]),
        "edgpu_sector_map": (C.c_int, [vp, i32, vp]),
        "edgpu_factor_size": (C.c_int, [vp, i32, i32, P(i64), P(i64)]),
        "edgpu_factor_export": (C.c_int, [vp, i32, i32, vp, vp, vp]),
        "edgpu_diag_export": (C.c_int, [vp, vp]),
        "edgpu_sector_hmat": (C.c_int, [vp, vp]),
        "edgpu_csr_rows": (C.c_int, [vp, i64, i64, vp, vp, vp, i64]),
        "edgpu_hxv": (C.c_int, [vp, P(i32), vp, vp]),
        "edgpu_hxv64": (C.c_int, [vp, i64, vp, vp]),
        "edgpu_hxv_device": (C.c_int, [vp, vp, vp]),
        "edgpu_hxv_time": (C.c_int, [vp, vp, vp, i32, i32, P(C.c_float)]),
        "edgpu_lanczos_gs": (C.c_int, [vp, i32, dbl, i32, P(dbl), vp, P(i32)]),
        "edgpu_lanczos_gs_device": (C.c_int, [vp, i32, dbl, i32, P(dbl), vp, P(i32)]),
        "edgpu_lanczos_tridiag": (C.c_int, [vp, vp, i32, dbl, vp, vp, P(i32)]),
        "edgpu_lanczos_tridiag_device": (C.c_int, [vp, vp, i32, dbl, vp, vp, P(i32)]),
        "edgpu_lanczos_time": (C.c_int, [vp, i32, i32, P(C.c_float)]),
        "edgpu_lanczos_bench_begin": (C.c_int, [vp, i32]),
        "edgpu_lanczos_bench_steps": (C.c_int, [vp, i32, i32]),
        "edgpu_lanczos_bench_comm_ms": (C.c_int, [vp, i32, P(C.c_float)]),
        "edgpu_lanczos_bench_result": (C.c_int, [vp, i32, P(C.c_float), vp, vp, i32]),
        "edgpu_eigsh": (C.c_int, [vp, i32, i32, i32, dbl, vp, vp, P(i32), P(i32), P(i32)]),
        "edgpu_dvdson": (C.c_int, [vp, i32, i32, dbl, vp, vp, P(i32), P(i32)]),
        "edgpu_eigsh_ncv_cap": (i32, []),
        "edgpu_eigsh_ncv_used": (i32, [vp]),
        "edgpu_lanczos_is_fused": (i32, [vp]),
        "edgpu_sector_describe": (C.c_int, [vp, vp, i32]),
        "edgpu_cache_release": (C.c_int, []),
        "edgpu_imp_histogram": (C.c_int, [vp, vp, i32, vp]),
        "edgpu_local_energy": (C.c_int, [vp, vp, vp]),
        "edgpu_twin_reorder": (C.c_int, [vp, vp, vp]),
        "edgpu_imp_density_offdiag": (C.c_int, [vp, vp, vp]),
        "edgpu_apply_op": (C.c_int, [vp, vp, i32, i32, i32, vp, vp, P(dbl)]),
        "edgpu_gf_lanczos": (C.c_int, [vp, vp, i32, i32, i32, vp, i32, dbl, P(dbl), vp, vp, P(i32)]),
        "edgpu_chi_lanczos": (C.c_int, [vp, vp, vp, vp, i32, dbl, P(dbl), vp, vp, P(i32)]),
        "edgpu_phonon_lanczos": (C.c_int, [vp, vp, i32, dbl, P(dbl), vp, vp, P(i32)]),
        "edgpu_sp_init_matrix": (C.c_int, [i64, i64, P(vp)]),
        "edgpu_sp_init_matrix_mpi": (C.c_int, [i64, i64, i64, i64, i64, P(vp)]),
        "edgpu_sp_load_matrix": (C.c_int, [vp, i64, i64, P(vp)]),
        "edgpu_sp_insert_element": (C.c_int, [vp, dbl, i64, i64]),
        "edgpu_sp_delete_matrix": (C.c_int, [vp]),
        "edgpu_sp_dump_matrix": (C.c_int, [vp, vp]),
        "edgpu_sp_row": (C.c_int, [vp, i64, P(i32), vp, vp, i32]),
        "edgpu_sp_matvec": (C.c_int, [vp, vp, vp, i32]),
        "edgpu_sp_matvec_device": (C.c_int, [vp, vp, vp, i32, vp]),
        "edgpu_comm_unique_id": (C.c_int, [vp]),
        "edgpu_comm_init_rccl": (C.c_int, [vp, vp]),
        "edgpu_comm_init_callbacks": (C.c_int, [vp, ALLGATHERV_FN, EXCHANGE_FN, ALLREDUCE_FN, vp]),
        "edgpu_comm_create_rccl": (C.c_int, [vp, i32, i32, i32, P(vp)]),
        "edgpu_comm_create_callbacks": (C.c_int, [i32, i32, ALLGATHERV_FN, EXCHANGE_FN, ALLREDUCE_FN, vp, P(vp)]),
        "edgpu_comm_create_emulated": (C.c_int, [i32, i32, P(vp)]),
        "edgpu_comm_sub": (C.c_int, [vp, i32, P(vp), P(i32)]),
        "edgpu_comm_destroy": (C.c_int, [vp]),
        "edgpu_sector_attach_comm": (C.c_int, [vp, vp]),
        "edgpu_comm_count": (C.c_int, [P(i32)]),
        "edgpu_comm_allreduce_sum": (C.c_int, [vp, vp, i32]),
        "edgpu_sector_exchange_info": (C.c_int, [vp, P(i32), vp, vp]),
        "edgpu_sector_relay_info": (C.c_int, [vp, P(i32), P(i32), P(i64), P(i64), vp, vp]),
        "edgpu_device_count": (C.c_int, [P(i32)]),
        "edgpu_malloc": (C.c_int, [P(vp), C.c_size_t]),
        "edgpu_free": (C.c_int, [vp]),
        "edgpu_memcpy_h2d": (C.c_int, [vp, vp, C.c_size_t]),
        "edgpu_memcpy_d2h": (C.c_int, [vp, vp, C.c_size_t]),
        "edgpu_memset": (C.c_int, [vp, i32, C.c_size_t]),
        "edgpu_device_synchronize": (C.c_int, []),
        "edgpu_last_error": (C.c_char_p, []),
        "edgpu_abi_version": (C.c_int, []),
        "edgpu_knob_count": (C.c_int, []),
        "edgpu_knob_info": (C.c_int, [i32, P(C.c_char_p), P(i32), P(C.c_char_p), P(C.c_char_p), P(C.c_char_p)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def knobs():
    """the environment switches of the loaded library: list of dicts (name, default, kind, doc, covered_by)"""
    L = lib()
    out = []
    for i in range(L.edgpu_knob_count()):
        n, k, d, t = C.c_char_p(), C.c_char_p(), C.c_char_p(), C.c_char_p()
        dv = C.c_int32()
        if L.edgpu_knob_info(i, C.byref(n), C.byref(dv), C.byref(k), C.byref(d), C.byref(t)) != 0:
            raise EdgpuError(-1, L.edgpu_last_error().decode())
        out.append({"name": n.value.decode(), "default": dv.value, "kind": k.value.decode(), "doc": d.value.decode(), "covered_by": t.value.decode()})
    return out


def check(rc):
    if rc != 0:
        raise EdgpuError(rc, lib().edgpu_last_error().decode())
