"""ctypes binding of libtnml.so (include/tnml.h) -- the only way Python reaches the HIP path.

`import torch` happens first on purpose: the PyTorch wheel bundles its own ROCm runtime
(libamdhip64.so.7, librccl.so.1, librocblas.so.5, librocsolver.so.0 -- same SONAMEs as
/opt/rocm/lib), and a process must never hold two HIP runtimes.  Loading torch first makes
libtnml.so bind to the copies torch already mapped.  The library fails loudly when it is missing
or when no HIP device is usable: there is no CPU fallback on the product path.
"""
import ctypes as C
import os

import numpy as np
import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

NL = 10
MAX_PASS = 64
DTYPES = {"f32": 0, "f64_e32": 1, "f64": 2, "bf16": 3, "bf16x3": 4}    # TNML_F32 / TNML_F64_E32 / TNML_F64 / TNML_BF16 / TNML_BF16X3 (include/tnml.h)
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtnml.so")

EXPORTS = [
    "tnml_create", "tnml_destroy", "tnml_last_error", "tnml_comm_unique_id", "tnml_comm_init",
    "tnml_set_data_u8", "tnml_set_data_phi", "tnml_set_site", "tnml_site_dims", "tnml_get_site",
    "tnml_env_init", "tnml_set_bond", "tnml_shift_env", "tnml_env_dims", "tnml_get_env",
    "tnml_bond_dims", "tnml_bond_tensor", "tnml_forward", "tnml_gradient", "tnml_quadcost",
    "tnml_cgrad", "tnml_svd_split", "tnml_bond_update", "tnml_truncate", "tnml_sweepnext",
    "tnml_shard_bounds", "tnml_profile_enable", "tnml_profile_select", "tnml_profile_count", "tnml_profile_get",
    "tnml_profile_reset", "tnml_synchronize", "tnml_device_bytes", "tnml_svd_stats", "tnml_classify", "tnml_replica_check",
    "tnml_estimate_bytes", "tnml_device_memory", "tnml_plan_maxm", "tnml_set_option", "tnml_comm_init_local", "tnml_comm_init_oneshot", "tnml_collective_mode", "tnml_bond_update_begin", "tnml_bond_update_end", "tnml_replica_repairs", "tnml_pAp", "tnml_collective_stats", "tnml_last_warning",
    "tnml_exact", "tnml_set_option_real", "tnml_pinv", "tnml_env_stats", "tnml_oneshot_export", "tnml_oneshot_connect", "tnml_oneshot_mem_kind", "tnml_split_stats", "tnml_oneshot_region_bytes",
]


class Config(C.Structure):
    _fields_ = [("device", C.c_int), ("rank", C.c_int), ("nranks", C.c_int), ("N", C.c_int),
                ("NT_local", C.c_int), ("NT_total", C.c_int64), ("maxm", C.c_int), ("dtype", C.c_int),
                ("svd_backend", C.c_int), ("mode", C.c_int), ("target_label", C.c_int)]


class CgTrace(C.Structure):
    _fields_ = [("npass_done", C.c_int), ("converged", C.c_int), ("cost", C.c_double * MAX_PASS),
                ("rnorm", C.c_double * MAX_PASS), ("pAp", C.c_double * MAX_PASS), ("alpha", C.c_double * MAX_PASS)]


class SweepParams(C.Structure):
    _fields_ = [("maxm", C.c_int), ("minm", C.c_int), ("cutoff", C.c_double), ("npass", C.c_int),
                ("lambda_", C.c_double), ("lambda_cost", C.c_double), ("cconv", C.c_double), ("report_costs", C.c_int)]


class BondReport(C.Structure):
    _fields_ = [("bond", C.c_int), ("half", C.c_int), ("c", C.c_int), ("mL", C.c_int), ("mR", C.c_int),
                ("label_on_B", C.c_int), ("origm", C.c_int), ("newm", C.c_int),
                ("truncerr", C.c_double), ("norm_newB", C.c_double), ("diff_B_newB", C.c_double),
                ("cost_after_svd", C.c_double), ("label_cost", C.c_double * NL), ("reg_cost", C.c_double),
                ("ncorrect", C.c_int64), ("cg", CgTrace), ("cost_old", C.c_double), ("cost_cg", C.c_double),
                ("reg_cost_cg", C.c_double), ("norm_oB", C.c_double)]


_lib = None


def load():
    """Load libtnml.so; raises (never falls back) when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C tnml_amd/csrc` (the HIP extension is mandatory, there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    L.tnml_create.argtypes = [C.POINTER(vp), C.POINTER(Config)]
    L.tnml_destroy.argtypes = [vp]
    L.tnml_last_error.restype = C.c_char_p
    L.tnml_last_error.argtypes = [vp]
    L.tnml_comm_unique_id.argtypes = [vp]
    L.tnml_comm_init.argtypes = [vp, vp]
    L.tnml_set_data_u8.argtypes = [vp, C.POINTER(C.c_uint8), C.POINTER(C.c_int32)]
    L.tnml_set_data_phi.argtypes = [vp, dp, C.POINTER(C.c_int32)]
    L.tnml_set_site.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    L.tnml_site_dims.argtypes = [vp, C.c_int, ip, ip, ip]
    L.tnml_get_site.argtypes = [vp, C.c_int, dp]
    L.tnml_env_init.argtypes = [vp]
    L.tnml_set_bond.argtypes = [vp, C.c_int]
    L.tnml_shift_env.argtypes = [vp, C.c_int, C.c_int]
    L.tnml_env_dims.argtypes = [vp, C.c_int, ip, ip]
    L.tnml_get_env.argtypes = [vp, C.c_int, dp]
    L.tnml_env_stats.argtypes = [vp] + [C.POINTER(C.c_int64)] * 4
    L.tnml_bond_dims.argtypes = [vp, C.c_int, ip, ip, ip]
    L.tnml_bond_tensor.argtypes = [vp, C.c_int, dp]
    L.tnml_forward.argtypes = [vp, dp, dp]
    L.tnml_gradient.argtypes = [vp, dp, dp]
    L.tnml_quadcost.argtypes = [vp, dp, C.c_double, dp, dp, dp, C.POINTER(C.c_int64)]
    L.tnml_cgrad.argtypes = [vp, dp, C.c_int, C.c_double, C.c_double, C.POINTER(CgTrace)]
    L.tnml_exact.argtypes = [vp, dp, C.c_double, C.c_double]
    L.tnml_pinv.argtypes = [vp, dp, C.c_int, C.c_int, C.c_double, C.c_double, dp, dp, C.POINTER(C.c_int), dp]
    L.tnml_set_option_real.argtypes = [vp, C.c_char_p, C.c_double]
    L.tnml_svd_split.argtypes = [vp, dp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, dp, ip, dp, ip]
    L.tnml_bond_update.argtypes = [vp, C.c_int, C.c_int, C.POINTER(SweepParams), C.POINTER(BondReport)]
    L.tnml_bond_update_begin.argtypes = [vp, C.c_int, C.c_int, C.POINTER(SweepParams)]
    L.tnml_bond_update_end.argtypes = [vp, C.POINTER(BondReport)]
    L.tnml_truncate.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_double, dp]
    L.tnml_sweepnext.argtypes = [ip, ip, C.c_int]
    L.tnml_sweepnext.restype = None
    L.tnml_shard_bounds.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.tnml_shard_bounds.restype = None
    L.tnml_profile_enable.argtypes = [vp, C.c_int]
    L.tnml_profile_select.argtypes = [vp, C.c_char_p]
    L.tnml_profile_count.argtypes = [vp]
    L.tnml_profile_get.argtypes = [vp, C.c_int, C.c_char_p, C.POINTER(C.c_int64), dp]
    L.tnml_profile_reset.argtypes = [vp]
    L.tnml_synchronize.argtypes = [vp]
    L.tnml_svd_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), dp, dp]
    L.tnml_device_bytes.argtypes = [vp]
    L.tnml_classify.argtypes = [vp, dp, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.tnml_device_bytes.restype = C.c_int64
    L.tnml_replica_check.argtypes = [vp, ip]
    L.tnml_replica_repairs.argtypes = [vp]
    L.tnml_replica_repairs.restype = C.c_int64
    L.tnml_collective_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.tnml_pAp.argtypes = [vp, dp, C.c_double, dp]
    L.tnml_last_warning.argtypes = [vp]
    L.tnml_last_warning.restype = C.c_char_p
    L.tnml_comm_init_local.argtypes = [C.POINTER(vp), C.c_int]
    L.tnml_comm_init_oneshot.argtypes = [C.POINTER(vp), C.c_int]
    L.tnml_collective_mode.argtypes = [vp]
    L.tnml_oneshot_export.argtypes = [vp, C.c_char_p]
    L.tnml_oneshot_connect.argtypes = [vp, C.c_char_p]
    L.tnml_oneshot_mem_kind.argtypes = [vp]
    L.tnml_split_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    L.tnml_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.tnml_estimate_bytes.argtypes = [C.POINTER(Config)]
    L.tnml_estimate_bytes.restype = C.c_int64
    L.tnml_oneshot_region_bytes.argtypes = [C.POINTER(Config)]
    L.tnml_oneshot_region_bytes.restype = C.c_int64
    L.tnml_device_memory.argtypes = [C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.tnml_plan_maxm.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_int64]
    _lib = L
    return L


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def flat(a):
    """fp64 copy flattened in ITensor (first-index-fastest) order"""
    return np.array(np.asarray(a, dtype=np.float64).ravel(order="F"), copy=True)


def truncate(p, maxm, minm, cutoff):
    p = np.ascontiguousarray(p, dtype=np.float64)
    te = C.c_double()
    m = load().tnml_truncate(dptr(p), len(p), maxm, minm, cutoff, C.byref(te))
    return m, te.value


def sweepnext(b, ha, N):
    bb, hh = C.c_int(b), C.c_int(ha)
    load().tnml_sweepnext(C.byref(bb), C.byref(hh), N)
    return bb.value, hh.value


def plan_maxm(N, NT_local, wanted, floor_m=1, dtype="f64", device=None, single=False, budget_bytes=None):
    """largest bond dimension <= wanted that an N-site MPS can reach and whose context fits the device (or budget_bytes)"""
    L = load()
    cfg = Config(device or 0, 0, 1, N, NT_local, NT_local, wanted, DTYPES[dtype], 0, 1 if single else 0, 0)
    if budget_bytes is None:
        budget_bytes = 0
        if device is not None:
            f, t = C.c_int64(), C.c_int64()
            if L.tnml_device_memory(device, C.byref(f), C.byref(t)) == 0:
                budget_bytes = int(f.value * 0.97)
    return L.tnml_plan_maxm(C.byref(cfg), wanted, floor_m, budget_bytes)


def estimate_bytes(N, NT_local, maxm, dtype="f64", single=False):
    cfg = Config(0, 0, 1, N, NT_local, NT_local, maxm, DTYPES[dtype], 0, 1 if single else 0, 0)
    return load().tnml_estimate_bytes(C.byref(cfg))


def shard_bounds(NT_total, nranks, rank):
    b, e = C.c_int64(), C.c_int64()
    load().tnml_shard_bounds(NT_total, nranks, rank, C.byref(b), C.byref(e))
    return b.value, e.value
