"""ctypes loader of libpinfmax_hip.so (the C ABI declared in include/pinfmax.h).

There is no Python or CPU implementation behind this module: if the shared
object is missing it raises, and pf_create itself fails without a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
# PINFMAX_LIB: another build of the same library (A/B measurements of a kernel variant, csrc/Makefile VARIANT=...)
SO = os.environ.get("PINFMAX_LIB") or os.path.join(PKG, "libpinfmax_hip.so")
NBINS = 210
MAX_SMOOTH = 64

FLAG_TIMING = 1
FLAG_DOUBLE_PRODUCTS = 2


class Config(C.Structure):
    _fields_ = [("n", C.c_int64), ("rank", C.c_int), ("nranks", C.c_int), ("device", C.c_int),
                ("field_bytes", C.c_int), ("flags", C.c_int)]


class ProductLayout(C.Structure):
    _fields_ = [("stride", C.c_size_t), ("off_Rmax", C.c_int), ("off_Fmax", C.c_int), ("off_Vel", C.c_int),
                ("off_Vel_2LPT", C.c_int), ("off_Vel_3LPT_1", C.c_int), ("off_Vel_3LPT_2", C.c_int)]


class GenicParams(C.Structure):
    _fields_ = [("Omega0", C.c_double), ("OmegaBaryon", C.c_double), ("Hubble100", C.c_double),
                ("PrimordialIndex", C.c_double), ("BoxSize_true_Mpc", C.c_double), ("PkNorm", C.c_double),
                ("RandomSeed", C.c_uint), ("FixedIC", C.c_int), ("PairedIC", C.c_int),
                ("pk_n", C.c_int), ("pk_logk", C.POINTER(C.c_double)), ("pk_logk3p", C.POINTER(C.c_double)),
                ("spectrum", C.c_int), ("WDM_PartMass_in_kev", C.c_double), ("UnitLength_in_cm", C.c_double)]


class CpuTime(C.Structure):
    _fields_ = [("fmax", C.c_double), ("deriv", C.c_double), ("fft", C.c_double), ("coll", C.c_double),
                ("lpt", C.c_double), ("mem_transf", C.c_double)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double), ("alg_bytes", C.c_double)]


ALLTOALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t),
                           C.POINTER(C.c_size_t), C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)

# every symbol include/pinfmax.h declares: name -> (restype, argtypes)
_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
PROTOTYPES = {
    "pf_layout_3lpt": (None, [C.POINTER(ProductLayout)]),
    "pf_create": (C.c_int, [C.POINTER(_vp), C.POINTER(Config)]),
    "pf_destroy": (C.c_int, [_vp]),
    "pf_last_error": (C.c_char_p, []),
    "pf_set_exchange": (C.c_int, [_vp, ALLTOALL_FN, _vp]),
    "pf_set_exchange_rows": (C.c_int, [_vp, ALLTOALLV_FN, _vp]),
    "pf_rccl_available": (C.c_int, []),
    "pf_release_rccl": (C.c_int, [_vp]),
    "pf_rccl_comm_count": (C.c_int, [_vp]),
    "pf_rccl_unique_id": (C.c_int, [_vp]),
    "pf_rccl_version": (C.c_int, [C.POINTER(C.c_int)]),
    "pf_init_rccl": (C.c_int, [_vp, _vp]),
    "pf_set_allreduce": (C.c_int, [_vp, ALLREDUCE_FN, _vp]),
    "pf_fabric_create": (_vp, [C.c_int]),
    "pf_fabric_destroy": (None, [_vp]),
    "pf_fabric_attach": (C.c_int, [_vp, _vp]),
    "pf_fabric_set_delay": (C.c_int, [_vp, C.c_int]),
    "pf_debug_exchange": (C.c_int, [_vp, C.c_size_t]),
    "pf_exchange_buffers": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "pf_set_stream": (C.c_int, [_vp, _vp]),
    "pf_get_stream": (_vp, [_vp]),
    "pf_set_density": (C.c_int, [_vp, _dp]),
    "pf_synth_density": (C.c_int, [_vp, C.c_uint64, C.c_double, C.c_double]),
    "pf_pk_norm": (C.c_int, [C.POINTER(GenicParams), C.c_double, _dp]),
    "pf_genic_density": (C.c_int, [_vp, C.POINTER(GenicParams)]),
    "pf_set_invgrow": (C.c_int, [_vp, C.c_int, _dp, _dp, C.c_int]),
    "pf_set_growth": (C.c_int, [_vp, _dp]),
    "pf_sweep": (C.c_int, [_vp, C.c_int, _dp, _dp]),
    "pf_second_derivatives": (C.c_int, [_vp, C.c_double]),
    "pf_collapse_times": (C.c_int, [_vp, C.c_int, _dp]),
    "pf_displacements": (C.c_int, [_vp, C.c_int, C.c_int]),
    "pf_fmax_pdf": (C.c_int, [_vp, C.POINTER(C.c_ulonglong)]),
    "pf_get_products": (C.c_int, [_vp, _vp, C.POINTER(ProductLayout)]),
    "pf_derivative": (C.c_int, [_vp, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_double)]),
    "pf_select_sorted": (C.c_int, [_vp, C.c_float, C.c_size_t, C.POINTER(C.c_uint), C.POINTER(C.c_float), C.POINTER(C.c_size_t)]),
    "pf_get_block": (C.c_int, [_vp, C.c_char_p, C.c_int, _vp]),
    "pf_set_collapse_model": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double)]),
    "pf_set_modified_gravity": (C.c_int, [_vp, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double)]),
    "pf_set_tabulated_ct": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double)]),
    "pf_ct_build": (C.c_int, [_vp, C.c_int, C.c_double, C.POINTER(C.c_double)]),
    "pf_ct_load": (C.c_int, [_vp, C.c_int, C.c_double, C.POINTER(C.c_double)]),
    "pf_debug_math": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_double)]),
    "pf_debug_stream_rate": (C.c_int, [_vp, C.c_int, C.c_int, _dp]),
    "pf_debug_lines": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _dp, _dp]),
    "pf_debug_pk": (C.c_int, [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]),
    "pf_debug_gfft": (C.c_int, [C.c_int, C.c_int, _dp, _dp]),
    "pf_debug_gfft_lines": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp]),
    "pf_debug_strided_jobs": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, _dp, _dp]),
    "pf_debug_invariant_reruns": (C.c_int, [_vp]),
    "pf_solve_ran_beside_zpass": (C.c_int, [_vp]),
    "pf_transform_path": (C.c_int, [_vp]),
    "pf_invgrow_table_status": (C.c_int, [_vp, C.c_int, _dp]),
    "pf_set_loopback_exchange": (C.c_int, [_vp, C.c_int]),
    "pf_loopback_active": (C.c_int, [_vp]),
    "pf_set_sources_in_sweep": (C.c_int, [_vp, C.c_int]),
    "pf_set_lpt_order": (C.c_int, [_vp, C.c_int]),
    "pf_set_ct_interpolation": (C.c_int, [_vp, C.c_int]),
    "pf_set_transposed_spectra": (C.c_int, [_vp, C.c_int]),
    "pf_replicated_spectrum": (C.c_int, [_vp]),
    "pf_update_products": (C.c_int, [_vp, _vp, C.POINTER(ProductLayout)]),
    "pf_set_growth_table": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_double, C.c_double, C.c_double]),
    "pf_get_second_derivative": (C.c_int, [_vp, C.c_int, _dp]),
    "pf_get_kvector": (C.c_int, [_vp, C.c_int, _dp]),
    "pf_get_density": (C.c_int, [_vp, _dp]),
    "pf_debug_replicated_rows": (C.c_int, [_vp, C.c_int, C.c_int, _dp]),
    "pf_plan_bytes": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "pf_forward_transform": (C.c_int, [_vp, _dp, _dp]),
    "pf_reverse_transform": (C.c_int, [_vp, _dp, _dp]),
    "pf_collapse_cells": (C.c_int, [_vp, C.c_int, _dp, C.c_size_t, _dp]),
    "pf_get_cputime": (C.c_int, [_vp, C.POINTER(CpuTime)]),
    "pf_reset_cputime": (C.c_int, [_vp]),
    "pf_kernel_stats": (C.c_int, [_vp, C.POINTER(KernelStat), C.c_int, C.POINTER(C.c_int)]),
    "pf_reset_kernel_stats": (C.c_int, [_vp]),
    "pf_synchronize": (C.c_int, [_vp]),
    "pf_device_bytes": (C.c_size_t, [_vp]),
}

_lib = None


def source_sha() -> str:
    """16 hex digits identifying the kernel sources next to this file (csrc/*.hip, *.h, *.cpp): stamps counter profiles, so
    that numbers measured on other kernels are never attached to a run (bench.py, profiles/tools/summarise.py)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.h")) +
                       glob.glob(os.path.join(PKG, "csrc", "*.cpp"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load():
    """Load the HIP library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise ImportError(
                f"{SO} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). pinocchio_amd has no CPU fallback.")
        L = C.CDLL(SO)
        for name, (res, args) in PROTOTYPES.items():
            f = getattr(L, name)  # AttributeError if the ABI and the header disagree
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib
