"""ctypes binding of librecman_hip.so (the C ABI declared in include/recman_hip.h).

There is NO fallback: if the library is missing or a symbol does not resolve the
import raises, and every call that returns a non-zero code raises RecmanHipError
with the library's thread-local message.  The product path never computes on the CPU.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# RECMAN_HIP_LIB: load another build of the same library (kernel experiments); default in-tree
LIB_PATH = os.environ.get("RECMAN_HIP_LIB") or os.path.join(_HERE, "csrc", "librecman_hip.so")


class RecmanHipError(RuntimeError):
    pass


P = c_void_p  # every device pointer and the stream travel as void*
I64 = c_int64

class MlpTail(ctypes.Structure):
    """rm_mlp_tail of include/recman_hip.h (the fused training head of the skinny MLP)."""
    _fields_ = [("logit_a", P), ("coef_a", c_float), ("logit_b", P), ("coef_b", c_float),
                ("coef_mlp", c_float), ("y", P), ("y_f", P), ("task", c_int), ("grad_scale", c_float),
                ("logit", P), ("pred", P), ("dlogit", P), ("loss_partial", P), ("loss", P),
                ("dh", P * 3)]


# name -> argtypes, in the order of include/recman_hip.h
SIGNATURES = {
    "rm_version": [],
    "rm_device_cus": [],
    "rm_profile_marker": [c_int, P],
    "rm_embed_fwd": [P, P, I64, P, P, I64, P, I64, P, P, P, P, c_int, P, P, I64, c_int, c_int,
                     P, P, P, P, c_int, P],
    "rm_linear_fwd": [P, P, P, P, P, P, I64, c_int, c_int, P, P],
    "rm_embed_bwd": [P, P, P, P, P, P, I64, c_int, c_int, P, P, P],
    "rm_scatter_add_rows": [P, P, P, P, I64, c_int, c_int, I64, P, P],
    "rm_linear_dense_bwd": [P, P, I64, c_int, P, P, P, P],
    "rm_logit_loss": [P, c_float, P, c_float, P, c_float, P, c_float, P, P, c_int, I64, P, P, P,
                      P, P, P],
    "rm_mlp_supported": [c_int, c_int, c_int, P],
    "rm_mlp_fwd": [P, P, c_int, c_int, c_int, P, P, P, P, P, c_int, I64, P, P, P, P],
    "rm_embed_mlp_fwd_supported": [c_int, c_int, I64, c_int, c_int, P],
    "rm_embed_mlp_fwd": [P, P, I64, P, c_int, c_int, P, P, P, c_int, I64, c_int, c_int, P, P, P, P, c_int,
                         c_int, P, P, P, P, P, c_int, P, P, P, P],
    "rm_mlp_bwd": [P, P, c_int, c_int, c_int, P, P, P, c_int, I64, P, P, P, c_int, P, P, P, P, P, P,
                   P, P, P, P, c_int, P],
    "rm_deepfm_step_supported": [c_int, c_int, I64, c_int, c_int, P],
    "rm_deepfm_step": [P, P, I64, P, P, c_int, P, P, I64, c_int, c_int, c_int, P, P, P, P, P, P, P, c_int, c_int,
                       c_float, P, P, P, P, P, P, P, P, P, P, P, P, I64, P, c_int, P],
    "rm_bias_act": [P, P, I64, c_int, c_int, P],
    "rm_act_bwd": [P, P, I64, c_int, c_int, P],
    "rm_outer_actgrad": [P, P, P, I64, c_int, c_int, P, P],
    "rm_outer_actgrad_sums": [P, P, P, I64, c_int, c_int, P, P, P, P, P, P],
    "rm_rowdot": [P, P, P, I64, c_int, P, P],
    "rm_cross_fwd": [P, P, c_int, c_int, P, P, P, c_int, I64, P, P, c_int, P],
    "rm_cross_bwd": [c_int, c_int, P, P, P, c_int, I64, P, P, c_int, P, P, P, P],
    "rm_cross_param_grads": [P, P, P, P, P, c_int, c_int, P, P, P, P],
    "rm_cin_layer_fwd": [P, P, I64, P, P, c_int, I64, c_int, c_int, c_int, c_int, P, P, c_int, c_int,
                         c_int, P, P],
    "rm_cin_layer_fwd6": [P, P, I64, P, P, c_int, I64, c_int, c_int, c_int, c_int, P, P, c_int, c_int,
                         c_int, P, P],
    "rm_cin_layer_bwd": [P, P, I64, c_int, P, c_int, P, P, I64, P, P, c_int, I64, c_int, c_int, c_int,
                         c_int, P, c_int, P, I64, P, P, P, I64, P],
    "rm_pool_rows": [P, I64, c_int, c_int, P, P, P, I64, P, P],
    "rm_pool_rows_bwd": [P, I64, P, P, c_int, P, P, P, I64, I64, P, P, P, P],
    "rm_pool_rows_padded": [P, c_int, c_int, P, I64, P, I64, P, I64, I64, c_int, P, P],
    "rm_pack_pooled_grad_rows": [P, I64, P, P, c_int, P, I64, P, I64, P, I64, I64, c_int, c_int, P, P],
    "rm_sparse_optimizer_prepare": [P, P, P, I64, c_int, I64, I64, P, I64, P],
    "rm_sparse_optimizer_step": [P, P, P, P, P, I64, c_int, c_int, I64, P, I64, P, c_int, c_int,
                                 c_float, c_float, c_float, c_float, c_int, c_float, c_float, P, I64, c_int, P, I64, P],
    "rm_sparse_optimizer_step_rows": [P, P, I64, I64, c_int, I64, P, I64, P, c_int, c_int,
                                      c_float, c_float, c_float, c_float, c_int, c_float, c_float, c_int, P, I64, P],
    "rm_dense_optimizer_step": [P, P, P, P, I64, c_int, c_int, c_float, c_float, c_float, c_float, c_int, P],
    "rm_shard_route": [P, P, I64, c_int, c_int, P, P, P, P, P],
    "rm_shard_route_padded": [P, P, I64, c_int, c_int, I64, P, P, P, P, P, P],
    "rm_pack_grad_rows": [P, P, P, P, P, I64, c_int, c_int, c_int, P, P],
    "rm_gather_rows": [P, I64, P, I64, c_int, P, P],
    "rm_permute_rows": [P, P, I64, c_int, c_int, P, P],
    "rm_dense_fwd": [P, I64, c_int, P, I64, c_int, P, I64, c_int, c_int, P, c_int, c_int, P, I64, P, I64,
                     I64, P, I64, P, I64, P, P],
    "rm_dense_fwd6": [P, I64, c_int, P, I64, c_int, P, I64, c_int, c_int, P, c_int, c_int, P, I64, I64, P, I64,
                      P, P, P, P, P],
    "rm_dense_wgrad": [P, I64, c_int, P, I64, c_int, P, I64, c_int, I64, P, I64, c_int, P, P, I64, P],
    "rm_dense_wgrad6": [P, I64, c_int, P, I64, c_int, P, I64, c_int, P, I64, c_int, I64, P, I64, P, I64, c_int, P, P,
                        I64, P],
}


# int64-returning size queries
SIGNATURES_I64 = {
    "rm_cin_filter_workspace": [c_int, c_int, c_int],
    "rm_cin_filter_workspace6": [c_int, c_int, c_int, c_int],
    "rm_cin_bwd_workspace": [I64, c_int, c_int, c_int, c_int],
    "rm_mlp_bwd_workspace": [c_int, c_int],
    "rm_deepfm_step_workspace": [c_int, c_int],
    "rm_outer_actgrad_sums_workspace": [I64, c_int],
    "rm_shard_route_workspace": [c_int],
    "rm_dense_filter_workspace": [c_int, c_int],
    "rm_dense6_workspace": [c_int, c_int, I64],
    "rm_dense_wgrad6_workspace": [c_int, c_int, I64],
    "rm_dense_wgrad_workspace": [c_int, c_int, I64],
    "rm_sparse_optimizer_workspace": [I64],
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise RecmanHipError(
            f"{LIB_PATH} not found: build it with `python -m recman_amd.build` "
            "(hipcc --offload-arch=gfx950). recman_amd has no CPU fallback.")
    # torch first: it bundles its own libamdhip64; loading ours before it would bring
    # in a second HIP runtime (the system one) that does not share torch's context
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH)
    lib.rm_last_error.restype = c_char_p
    lib.rm_last_error.argtypes = []
    for name, argtypes in SIGNATURES_I64.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int64
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud
        fn.argtypes = argtypes
        fn.restype = c_int
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def call(name, *args):
    """Calls an int-returning entry point; raises RecmanHipError on a non-zero code."""
    l = lib()
    rc = getattr(l, name)(*args)
    if rc != 0:
        msg = l.rm_last_error().decode("utf-8", "replace")
        raise RecmanHipError(f"{name} failed ({rc}): {msg}")
    return rc
