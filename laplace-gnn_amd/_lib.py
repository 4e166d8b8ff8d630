"""ctypes binding of ``liblaplace_gnn_hip.so`` (C ABI: include/laplace_gnn_hip.h).

The product path has NO fallback: if the library is missing or a call fails, an exception is
raised.  Nothing here imports the CPU oracle."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LGNN_LIB_DIR: developer A/B builds (make OUT_DIR=../lib_<variant>); the default is the in-tree lib/
LIB_PATH = os.path.join(_HERE, os.environ.get("LGNN_LIB_DIR", "lib"), "liblaplace_gnn_hip.so")

KIND_GCN, KIND_SAGE = 0, 1
ACT_RELU, ACT_TANH = 0, 1
LIK_CLASSIFICATION, LIK_REGRESSION = 0, 1
NORM_NONE, NORM_LAYER, NORM_BATCH = 0, 1, 2
FLAG_FORK_EXACT_SEED, FLAG_NO_FUSE, FLAG_NO_PATHS, FLAG_FORCE_PATHS = 1, 2, 4, 8

# name -> (restype, argtypes); mirrors include/laplace_gnn_hip.h one to one
_vp, _i64, _i32, _u32 = C.c_void_p, C.c_int64, C.c_int, C.c_uint32
_pp = C.POINTER(C.c_void_p)
SIGNATURES = {
    "lgnn_abi_version": (_i32, []),
    "lgnn_last_error": (C.c_char_p, []),
    "lgnn_create": (_i32, [_pp, _i64, _vp, _i64, _i32, _i32, _vp]),
    "lgnn_destroy": (None, [_vp]),
    "lgnn_nnz": (_i64, [_vp]),
    "lgnn_num_nodes": (_i64, [_vp]),
    "lgnn_is_symmetric": (_i32, [_vp]),
    "lgnn_num_long_rows": (_i64, [_vp]),
    "lgnn_kfac_last_route": (_i32, [_vp]),
    "lgnn_export_adj": (_i32, [_vp, _vp, _vp, _vp]),
    "lgnn_update_adjacency": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "lgnn_adj_to_edge_index": (_i32, [_vp, _vp, C.POINTER(_i64), _vp]),
    "lgnn_export_propagation": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "lgnn_bind_model": (_i32, [_vp, _i32, C.POINTER(_i64), _pp, _pp, _vp, _i32, _i32]),
    "lgnn_bind_extras": (_i32, [_vp, _pp, _pp, _i32, _pp, _pp, _pp, _pp, C.c_float]),
    "lgnn_invalidate": (_i32, [_vp]),
    "lgnn_device_bytes": (_i64, [_vp]),
    "lgnn_set_workspace_limit": (_i32, [_vp, _i64]),
    "lgnn_forward": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "lgnn_forward_all": (_i32, [_vp, _vp, _vp]),
    "lgnn_kfac_accumulate": (_i32, [_vp, _vp, _vp, _i64, _i64, _u32, _pp, _pp, _vp, _vp]),
    "lgnn_kfac_accumulate_classes": (_i32, [_vp, _vp, _vp, _i64, _i64, _u32, _i64, _i64, _pp, _pp, _vp, _vp]),
    "lgnn_kfac_accumulate_share": (_i32, [_vp, _vp, _vp, _i64, _i64, _u32, _i64, _i64, _i64, _pp, _pp, _vp, _vp]),
    "lgnn_kfac_accumulate_fisher": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _u32, C.c_float, C.c_float, _pp, _pp, _vp, _vp]),
    "lgnn_ef_accumulate": (_i32, [_vp, _vp, _vp, _vp, _i64, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "lgnn_kfac_plan": (_i32, [_i32, _i32, C.POINTER(_i64), _i64, _i64, _i32, _u32, _i64, C.POINTER(_i64)]),
    "lgnn_diag_accumulate": (_i32, [_vp, _vp, _vp, _i64, _u32, _vp, _vp, _vp]),
    "lgnn_full_accumulate": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "lgnn_lastlayer_full_accumulate": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "lgnn_lastlayer_features": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "lgnn_lastlayer_pairs_accumulate": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "lgnn_lastlayer_pairs_place": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "lgnn_check_async_errors": (_i32, [_vp, _vp]),
    "lgnn_peek_async_errors": (_i32, [_vp]),
    "lgnn_enable_kernel_timing": (_i32, [_vp, _i32]),
    "lgnn_kernel_timing_read": (_i32, [_vp, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(_i64)]),
    "lgnn_kernel_timing_launches": (_i32, [_vp, _vp, _i64, _vp]),
    "lgnn_kfac_adjgrad_batch": (_i32, [_vp, _vp, _vp, _i64, _u32, _pp, C.c_float, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "lgnn_adjgrad_finish": (_i32, [_vp, _vp, _pp, C.c_float, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "lgnn_diag_adjgrad_batch": (_i32, [_vp, _vp, _vp, _i64, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "lgnn_diag_adjgrad_finish": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "lgnn_glm_variance": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lgnn_glm_variance_mapped": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lgnn_symeig_batched": (_i32, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "lgnn_jacobians": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises HipLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C laplace-gnn_amd/csrc`).  There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export what the header declares
        fn.restype, fn.argtypes = res, args
    if lib.lgnn_abi_version() != 1:
        raise HipLibraryError("ABI version mismatch between _lib.py and liblaplace_gnn_hip.so")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().lgnn_last_error().decode("utf-8", "replace")
        raise HipLibraryError(f"{what} failed (rc={rc}): {msg}")


def ptr_array(ptrs):
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    return arr
