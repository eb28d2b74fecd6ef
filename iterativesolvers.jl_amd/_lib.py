"""ctypes binding of libmik.so -- the C ABI declared in include/mik.h.

The product path has no CPU fallback: if the HIP library is missing or a call fails, this module
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmik.so")

MIK_OK = 0
MIK_F64, MIK_F32 = 0, 1
MIK_MGS, MIK_CGS, MIK_DGKS = 0, 1, 2
STATUS = {1: "invalid argument", 2: "HIP runtime error", 3: "dimension/dtype mismatch",
          4: "out of memory", 5: "not implemented", 6: "callback failed", 7: "norm outside the safely representable range"}


class MikError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        super().__init__(f"{where}: status {code} ({STATUS.get(code, '?')}) {detail}".rstrip())


_vp = C.c_void_p
_i64 = C.c_int64
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

# callbacks of a row-partitioned iterable (include/mik.h: mik_halo_fn, mik_reduce_fn, mik_partition)
HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p)


# any operator / any preconditioner on the fused iterables (include/mik.h: mik_mul_fn, mik_ldiv_fn, mik_operator, mik_precond)
MUL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
LDIV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)


class MikOperator(C.Structure):
    _fields_ = [("dtype", C.c_int), ("n", C.c_int64), ("csr", C.c_void_p), ("mul", MUL_FN), ("user", C.c_void_p)]


class MikPrecond(C.Structure):
    _fields_ = [("diag", C.c_void_p), ("ldiv", LDIV_FN), ("user", C.c_void_p)]


class MikDeviceInfo(C.Structure):
    """include/mik.h mik_device_info"""
    _fields_ = [("device", C.c_int), ("compute_units", C.c_int), ("xcds", C.c_int), ("wavefront_size", C.c_int),
                ("lds_bytes_per_cu", C.c_int64), ("l2_bytes", C.c_int64), ("hbm_bytes", C.c_int64), ("arch", C.c_char * 64),
                ("planned_compute_units", C.c_int), ("planned_xcds", C.c_int), ("xcd_maps", C.c_int), ("resident_workgroup_cap", C.c_int),
                ("gs_single_launch_max_segments", C.c_int), ("gs_xcd_local_max_workgroups", C.c_int), ("sweep_grid_cap", C.c_int),
                ("mgs_resident_max_segments", C.c_int), ("reserved", C.c_int * 7)]


class MikPartition(C.Structure):
    _fields_ = [("rank", C.c_int), ("nranks", C.c_int), ("n_ext", C.c_int64), ("x_ext", C.c_void_p),
                ("send_idx", C.c_void_p), ("n_send", C.c_int64), ("send_buf", C.c_void_p),
                ("halo", HALO_FN), ("reduce", REDUCE_FN), ("user", C.c_void_p), ("link", C.c_void_p)]


# name -> (restype, argtypes); mirrors include/mik.h one to one
SIGNATURES = {
    "mik_abi_version": (C.c_int, []),
    "mik_device_count": (C.c_int, [_ip]),
    "mik_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "mik_ctx_destroy": (C.c_int, [_vp]),
    "mik_ctx_info": (C.c_int, [_vp, C.POINTER(MikDeviceInfo)]),
    "mik_dev_gmres_form": (C.c_int, [_vp, _ip, _ip, _ip, _ip]),
    "mik_dev_mgs_resident_shape": (C.c_int, [_ip, _ip, _ip]),
    "mik_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "mik_ctx_synchronize": (C.c_int, [_vp]),
    "mik_last_error": (C.c_char_p, [_vp]),
    "mik_reduce_shape": (C.c_int, [C.c_int, _ip, _ip]),
    "mik_spmv_dot_shape": (C.c_int, [_ip, _ip]),
    "mik_set_tuning": (C.c_int, [C.c_int, C.c_int]),
    "mik_ctx_set_tuning": (C.c_int, [_vp, C.c_int, C.c_int]),
    "mik_dev_xwin_plan": (C.c_int, [_i64, _ip, _ip, _ip, C.c_int, _i64, _i64, _ip, _ip]),
    "mik_spmv_long_row": (C.c_int, [_ip]),
    "mik_spmv_long_segment": (C.c_int, [_ip]),
    "mik_spmv_long_group": (C.c_int, [_ip]),
    "mik_malloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "mik_free": (C.c_int, [_vp, _vp]),
    "mik_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "mik_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "mik_copy": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp]),
    "mik_fill": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp]),
    "mik_csr_create": (C.c_int, [_vp, C.c_int, _i64, _i64, _i64, _i64p, _i64p, _vp, C.c_int, C.c_int,
                                 C.POINTER(_vp)]),
    "mik_csr_create_i32": (C.c_int, [_vp, C.c_int, _i64, _i64, _i64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _vp, C.c_int, C.c_int,
                                     C.POINTER(_vp)]),
    "mik_csr_destroy": (C.c_int, [_vp]),
    "mik_csr_layout": (C.c_int, [_vp, _ip]),
    "mik_csr_set_layout": (C.c_int, [_vp, C.c_int]),
    "mik_csr_stored_bytes": (C.c_int, [_vp, _i64p]),
    "mik_csr_compact": (C.c_int, [_vp]),
    "mik_spmv_kernel": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "mik_csr_info": (C.c_int, [_vp, _i64p, _i64p, _i64p, _ip]),
    "mik_spmv": (C.c_int, [_vp, _vp, _vp, _vp]),
    "mik_dot": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp, _vp]),
    "mik_nrm2": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp]),
    "mik_axpy": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp, _vp]),
    "mik_xpby": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp, _vp]),
    "mik_sub": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp]),
    "mik_scal": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp]),
    "mik_divide": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp, _vp]),
    "mik_orthogonalize": (C.c_int, [_vp, C.c_int, _i64, C.c_int, _vp, _i64, _vp, _vp, _vp, C.c_int]),
    "mik_orthogonalize_vectors": (C.c_int, [_vp, C.c_int, _i64, C.c_int, C.POINTER(_vp), _vp, _vp, _vp]),
    "mik_gemv_n": (C.c_int, [_vp, C.c_int, _i64, C.c_int, _vp, _i64, _vp, _vp, _vp]),
    "mik_gemv_t": (C.c_int, [_vp, C.c_int, _i64, C.c_int, _vp, _i64, _vp, _vp]),
    "mik_lu_solve": (C.c_int, [C.c_int, _vp, _i64, C.c_int, _vp]),
    "mik_cg_create": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, _i64,
                                C.c_int, C.POINTER(_vp)]),
    "mik_cg_create_op": (C.c_int, [_vp, C.POINTER(MikOperator), C.POINTER(MikPrecond), _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, _i64,
                                   C.c_int, C.POINTER(_vp)]),
    "mik_gmres_create_op": (C.c_int, [_vp, C.POINTER(MikOperator), C.POINTER(MikPrecond), C.POINTER(MikPrecond), _vp, _vp, C.c_double, C.c_double,
                                      C.c_int, _i64, C.c_int, C.c_int, C.POINTER(_vp)]),
    "mik_cg_destroy": (C.c_int, [_vp]),
    "mik_cg_fused_x": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "mik_cg_iterate": (C.c_int, [_vp, _i64, _f64p, _ip]),
    "mik_cg_iterate_many": (C.c_int, [_vp, _i64, _i64, _f64p, _i64p]),
    "mik_cg_state": (C.c_int, [_vp, _f64p, _f64p, _f64p, _i64p, _i64p, _ip]),
    "mik_gmres_create": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_int, _i64, C.c_int,
                                   C.c_int, C.POINTER(_vp)]),
    "mik_gmres_create_partitioned": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_int, _i64, C.c_int,
                                               C.c_int, C.POINTER(MikPartition), C.POINTER(_vp)]),
    "mik_gmres_destroy": (C.c_int, [_vp]),
    "mik_gmres_iterate": (C.c_int, [_vp, _i64, _f64p, _ip]),
    "mik_gmres_iterate_many": (C.c_int, [_vp, _i64, _i64, _f64p, _i64p]),
    "mik_gmres_state": (C.c_int, [_vp, _f64p, _f64p, _f64p, _ip, _i64p, _ip]),
    "mik_axpy_dot": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "mik_axpy2_nrm2": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "mik_cheb_direction": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp, _vp, C.c_int, _vp]),
    "mik_minres_update": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "mik_gram": (C.c_int, [_vp, C.c_int, _i64, C.c_int, _vp, _i64, _vp]),
    "mik_bicgstab_mr_update": (C.c_int, [_vp, C.c_int, _i64, C.c_int, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "mik_bicgstab_create": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _i64, _vp, _i64, _vp, _vp, C.POINTER(_vp)]),
    "mik_bicgstab_step": (C.c_int, [_vp, _vp]),
    "mik_bicgstab_destroy": (C.c_int, [_vp]),
    "mik_minres_create": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_int, C.POINTER(_vp)]),
    "mik_minres_step": (C.c_int, [_vp, _i64, _vp]),
    "mik_minres_proj_shape": (C.c_int, [_vp, _ip, _ip]),
    "mik_bicgstab_dot_shape": (C.c_int, [_vp, _ip, _ip]),
    "mik_minres_destroy": (C.c_int, [_vp]),
    "mik_xpby_nrm2": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp, _vp, _vp]),
    "mik_lsqr_update": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mik_lsmr_update": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mik_axpy2_dot": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mik_scal2": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp, _vp, _vp]),
    "mik_qmr_update": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mik_idrs_create": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, C.c_double, C.POINTER(_vp)]),
    "mik_idrs_step": (C.c_int, [_vp, C.c_int, _vp]),
    "mik_idrs_state": (C.c_int, [_vp, _vp, _vp, _vp]),
    "mik_idrs_destroy": (C.c_int, [_vp]),
    "mik_gather": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp, _vp]),
    "mik_cgd_create": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, C.c_int, C.c_int,
                                 C.c_double, C.c_double, _i64, C.c_int, C.POINTER(_vp)]),
    "mik_cgd_destroy": (C.c_int, [_vp]),
    "mik_cgd_phase": (C.c_int, [_vp, C.c_int, _i64]),
    "mik_cgd_set_interior": (C.c_int, [_vp, _i64, _i64]),
    "mik_cgd_wait": (C.c_int, [_vp, _f64p, _f64p, _ip, _f64p, _i64, _i64p]),
    "mik_cgd_set_halo_plan": (C.c_int, [_vp, C.c_int, _ip, _i64p, _i64p, C.c_int, _ip, _i64p, _i64p]),
    "mik_cgd_halo_early": (C.c_int, [_vp, _ip, _i64p, _ip]),
    "mik_comm_unique_id": (C.c_int, [_vp]),
    "mik_comm_create": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "mik_comm_destroy": (C.c_int, [_vp]),
    "mik_comm_info": (C.c_int, [_vp, _ip, _ip, _ip]),
    "mik_comm_allgather_sum": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "mik_comm_halo": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _ip, _i64p, _i64p, C.c_int, _ip, _i64p, _i64p]),
    "mik_cgd_set_comm": (C.c_int, [_vp, _vp]),
    "mik_comm_mailbox_export": (C.c_int, [_vp, _vp]),
    "mik_comm_mailbox_connect": (C.c_int, [_vp, _vp]),
    "mik_comm_mailbox_info": (C.c_int, [_vp, _ip, _ip]),
    "mik_cgd_ghost_export": (C.c_int, [_vp, _vp]),
    "mik_cgd_connect_ghosts": (C.c_int, [_vp, _vp, _i64p, _i64p]),
    "mik_plink_create": (C.c_int, [_vp, C.c_int, _i64, C.c_int, _ip, _i64p, _i64p, C.c_int, _ip, _i64p, _i64p, C.POINTER(_vp)]),
    "mik_plink_export": (C.c_int, [_vp, _vp]),
    "mik_plink_connect": (C.c_int, [_vp, _vp, _i64p, _i64p]),
    "mik_plink_info": (C.c_int, [_vp, _ip, _ip, _i64p]),
    "mik_plink_exchange": (C.c_int, [_vp, _vp, _vp]),
    "mik_plink_destroy": (C.c_int, [_vp]),
    "mik_cgd_init": (C.c_int, [_vp, _f64p, _f64p]),
    "mik_cgd_iterate_many": (C.c_int, [_vp, _i64, _i64, _f64p, _i64p]),
    "mik_cgd_group_init": (C.c_int, [C.POINTER(_vp), C.c_int, _f64p, _f64p]),
    "mik_cgd_group_iterate_many": (C.c_int, [C.POINTER(_vp), C.c_int, _i64, _i64, _f64p, _i64p]),
    "mik_cgd_group_release": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "mik_hessenberg_ldiv": (C.c_int, [C.c_int, _vp, _i64, C.c_int, _vp]),
    "mik_givens": (C.c_int, [C.c_int, _vp, _vp, _vp]),
    "mik_time_spmv": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, _f64p]),
    "mik_cg_profile": (C.c_int, [_vp, C.c_int, _f64p, _i64p]),
    "mik_cgd_profile": (C.c_int, [_vp, C.c_int, _f64p, _i64p]),
    "mik_cg_profile_kernels": (C.c_int, [_vp, _f64p, _i64p]),
}

_lib = None


def _share_hip_runtime():
    """One HIP runtime per process.  libmik.so needs the unversioned "libamdhip64.so" (csrc/Makefile).
    PyTorch-ROCm bundles its own copy under that name; if torch is installed, map that copy first
    (without importing torch) so that libmik.so and a later `import torch` both bind to it.  Two
    different runtimes in one process break whichever initialises second (observed: torch reports
    "no ROCm-capable device" after the system runtime has opened the GPU).  Without torch the name
    resolves through libmik.so's RUNPATH to /opt/rocm/lib."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("MIK_NO_TORCH_RUNTIME") == "1":
        return                                  # torch's runtime is already mapped (or sharing is declined)
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None:
        return
    # Importing torch maps its bundled runtime under the bare name "libamdhip64.so", which is what
    # libmik.so's DT_NEEDED then matches.  (dlopen() of the same file by absolute path does not
    # register that name, and the system runtime would be loaded beside it.)
    import torch  # noqa: F401


def lib() -> C.CDLL:
    """Load libmik.so (built in-tree by ``__graft_entry__.build()``).  Fails loudly if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
                "There is no CPU fallback for the product path.")
        _share_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        for kv in filter(None, os.environ.get("MIK_TUNING", "").split(",")):   # development knobs, e.g. "0=1"
            k, v = kv.split("=")
            L.mik_set_tuning(int(k), int(v))
        _lib = L
    return _lib


def dtype_code(dtype) -> int:
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return MIK_F64
    if dtype == np.float32:
        return MIK_F32
    raise TypeError(f"libmik supports float64 and float32 (complex is out of scope), got {dtype}")


def check(code: int, where: str, ctx_handle=None):
    if code != MIK_OK:
        detail = lib().mik_last_error(ctx_handle)
        raise MikError(code, where, detail.decode() if detail else "")
