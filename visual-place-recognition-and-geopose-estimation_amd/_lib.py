"""ctypes binding of libvpr_amd.so — one prototype per entry point of include/vpr_amd.h.

There is no fallback: if the library is missing, `lib()` raises with the build command.
"""
import ctypes
import os

import torch  # noqa: F401  MUST precede loading libvpr_amd.so: the library has to bind to the HIP
              # runtime PyTorch ships (torch/lib/libamdhip64.so), not to a second copy from /opt/rocm —
              # two runtimes in one process make every launch fail with hipErrorNoDevice.
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_longlong, c_size_t,
                    c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ABI_VERSION = 3

STATUS_OK = 0


class SaladWeightsC(Structure):
    """struct vpr_salad_weights (device pointers; the two *_frag members may be null)."""
    _fields_ = [(n, c_void_p) for n in
                ("w1_sc", "b1_sc", "w2_s", "b2_s", "w2_c", "b2_c", "w1_t", "b1_t", "w2_t", "b2_t", "w2_s_frag", "w2_c_frag")]


class SaladWeightsF32C(Structure):
    """struct vpr_salad_weights_f32 (device pointers)."""
    _fields_ = [(n, c_void_p) for n in
                ("w1_sc", "b1_sc", "w2_s", "b2_s", "w2_c", "b2_c", "w1_t", "b1_t", "w2_t", "b2_t")]


# name -> (restype, argtypes); kept in one table so tests can check every symbol is exported
PROTOTYPES = {
    "vpr_status_string": (c_char_p, [c_int]),
    "vpr_abi_version": (c_int, []),
    "vpr_tuning_set": (c_int, [c_char_p, c_int, c_int]),
    "vpr_tuning_get": (c_int, [c_char_p, POINTER(c_int)]),
    "vpr_salad_workspace_bytes": (c_size_t, [c_int] * 7),
    "vpr_salad_aggregate": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(SaladWeightsC), c_float,
                                    c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    "vpr_salad_aggregate_split": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(SaladWeightsC), c_float,
                                          c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                          c_void_p, c_size_t, c_void_p]),
    "vpr_salad_stage_token": (c_int, [c_void_p, c_longlong, c_int, c_int, c_int, POINTER(SaladWeightsC), c_int, c_int, c_int, c_int,
                                      c_void_p, c_size_t, c_void_p]),
    "vpr_salad_stage_mlps": (c_int, [c_void_p, c_longlong, c_int, c_int, c_int, POINTER(SaladWeightsC), c_int, c_int, c_int, c_int,
                                     c_void_p, c_size_t, c_void_p]),
    "vpr_salad_stage_aggregate": (c_int, [c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                          c_void_p, c_size_t, c_void_p]),
    "vpr_salad_f32_workspace_bytes": (c_size_t, [c_int] * 7),
    "vpr_salad_pack_w2_fragments": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "vpr_salad_aggregate_f32": (c_int, [c_void_p, c_longlong, c_void_p, c_longlong, c_int, c_int, c_int,
                                        POINTER(SaladWeightsF32C), c_float, c_int, c_int, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vpr_salad_sinkhorn_aggregate": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                             c_int, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "vpr_gemm_nt_bf16": (c_int, [c_void_p, c_int, c_int, c_longlong, c_void_p, c_int, c_void_p, c_int,
                                 c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vpr_gemm256_nt_bf16": (c_int, [c_void_p, c_int, c_int, c_longlong, c_void_p, c_int, c_void_p, c_int,
                                    c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vpr_knn_workspace_bytes": (c_size_t, [c_int] * 4),
    "vpr_knn_topk": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                             c_void_p, c_size_t, c_void_p]),
    "vpr_knn_topk_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vpr_knn_topk_checked": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_float, c_void_p, c_void_p, c_void_p]),
    "vpr_knn_select_checked": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                       c_void_p, c_size_t, c_float, c_void_p, c_void_p, c_void_p]),
    "vpr_knn_topk_fp8_checked": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                         c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_void_p, c_void_p, c_void_p]),
    "vpr_knn_topk_exhaustive": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vpr_quantize_fp8_rows": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p]),
    "vpr_knn_scores": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "vpr_knn_select": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                               c_void_p, c_size_t, c_void_p]),
    "vpr_knn_topk_scores_stage": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                          c_void_p, c_size_t, c_void_p]),
    "vpr_knn_topk_select_stage": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_void_p, c_void_p, c_void_p]),
    "vpr_knn_scores_kernel_name": (c_char_p, [c_int, c_int, c_int]),
    "vpr_knn_scores_ptr": (c_void_p, [c_void_p, c_int, c_int, c_int, c_int, POINTER(c_int)]),
    "vpr_topk_merge": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vpr_pose_head_workspace_bytes": (c_size_t, [c_int] * 4),
    "vpr_pose_head": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                              c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "vpr_ln_meanpool_head": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
                                     c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "vpr_preprocess_workspace_bytes": (c_size_t, [c_int] * 3),
    "vpr_preprocess_resize_normalize": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int,
                                                c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                                POINTER(c_float), POINTER(c_float),
                                                c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vpr_layernorm_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_longlong, c_int, c_void_p]),
    "vpr_bias_layernorm_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_longlong, c_int, c_void_p]),
    "vpr_attention_qkv_split_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_longlong, c_int, c_int, c_float, c_void_p]),
    "vpr_skinny_linear_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                       c_int, c_int, c_int, c_void_p]),
    "vpr_bias_layernorm_cls_linear_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_longlong,
                                                   c_int, c_longlong, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                   c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "vpr_skinny_linear_stats_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                             c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vpr_pose_head_pack_w1": (c_int, [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p]),
    "vpr_pose_head_split_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vpr_pose_head_fused_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vpr_pose_head_fused_counter_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vpr_pose_head_pack_w1_frag": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vpr_pose_head_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                    c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "vpr_pose_head_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                    c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "vpr_head_train_workspace_bytes": (c_size_t, [c_int] * 4),
    "vpr_head_train_state_floats": (c_longlong, [c_int] * 3),
    "vpr_head_train_step": (c_int, [c_void_p, c_longlong, c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_int,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                    c_double, c_double, c_double, c_double, c_double, c_int, c_double, c_void_p, c_void_p, c_size_t,
                                    c_void_p]),
    "vpr_head_train_epoch": (c_int, [c_void_p, c_longlong, c_void_p, c_int, c_int, c_void_p, c_longlong, c_int, c_int, c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_double, c_double, c_double, c_double, c_double, c_int, c_double, c_void_p, c_void_p, c_size_t,
                                     c_void_p]),
    "vpr_patchify_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vpr_add_layernorm_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p,
                                       c_longlong, c_int, c_void_p]),
    "vpr_attention_qkv_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "vpr_f32_to_bf16": (c_int, [c_void_p, c_void_p, c_longlong, c_void_p]),
}


def library_path() -> str:
    """The in-tree library; VPR_AMD_LIBRARY points an A/B run at another build of it (scripts/ab_libs.sh)."""
    return os.environ.get("VPR_AMD_LIBRARY") or os.path.join(_HERE, "libvpr_amd.so")


def lib() -> ctypes.CDLL:
    """Load (once) and return the HIP library; raises if it has not been built."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: the HIP extension is required (there is no CPU fallback). "
                "Build it with `python -c \"import __graft_entry__ as g; g.build()\"` "
                "or `make -C visual-place-recognition-and-geopose-estimation_amd/csrc`.")
        handle = ctypes.CDLL(path)
        for name, (restype, argtypes) in PROTOTYPES.items():
            fn = getattr(handle, name)   # AttributeError if a declared symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.vpr_abi_version() != ABI_VERSION:
            raise RuntimeError("libvpr_amd.so ABI version mismatch; rebuild the library")
        _LIB = handle
    return _LIB


def tuning_set(name: str, value=None) -> None:
    """Set (or, with value None, unset) one of the library's A/B switches.  The library reads the VPR_* environment
    variables once at load; afterwards this call is the only way to change them (scripts/, tests)."""
    st = lib().vpr_tuning_set(name.encode(), 0 if value is None else int(value), int(value is None))
    check(st, f"vpr_tuning_set({name})")


def tuning_get(name: str):
    v = c_int(0)
    st = lib().vpr_tuning_get(name.encode(), ctypes.byref(v))
    if st == 1:
        return None
    check(st, f"vpr_tuning_get({name})")
    return v.value


class tuning:
    """Context manager: `with _lib.tuning(VPR_KNN_VARIANT=6): ...` sets the switches and restores the old values."""

    def __init__(self, **switches):
        self.switches, self.old = switches, {}

    def __enter__(self):
        for k, v in self.switches.items():
            self.old[k] = tuning_get(k)
            tuning_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            tuning_set(k, v)
        return False


def check(status: int, what: str) -> None:
    if status != STATUS_OK:
        msg = lib().vpr_status_string(status).decode()
        raise RuntimeError(f"{what} failed: {msg} (status {status})")
