"""ctypes binding of libegotap_hip.so (include/egotap.h).  No CPU fallback: if the HIP library is
missing or fails to load, everything here raises."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

ERR_NAMES = {1: "EGOTAP_ERR_INVALID", 2: "EGOTAP_ERR_HIP", 3: "EGOTAP_ERR_UNBOUND", 4: "EGOTAP_ERR_WORKSPACE"}
NET_LIFT, NET_HM_POS, NET_HM_ROT = 0, 1, 2
F32, I64 = 0, 1
PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16": 2}      # egotap.h EGOTAP_PREC_*


class EgotapConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "struct_bytes", "n_joints_hm", "estimate_head", "hm_size", "hidden", "vit_dim", "vit_heads", "vit_layers",
        "patch", "pu_hidden")] + [("hm_blocks", C.c_int32 * 4)]      # zeros = resnet18's (2, 2, 2, 2)


class EgotapError(RuntimeError):
    pass


_lib = None

_PROTOS = {
    "egotap_abi_version": (C.c_int, []),
    "egotap_last_error": (C.c_char_p, []),
    "egotap_create": (C.c_int, [C.POINTER(EgotapConfig), C.POINTER(C.c_void_p)]),
    "egotap_destroy": (None, [C.c_void_p]),
    "egotap_bind_param": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_int64, C.c_int]),
    "egotap_unbound_count": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "egotap_lift_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]),
    "egotap_lift_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_lift_intermediate": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]),
    "egotap_lift_debug_stop": (C.c_int, [C.c_void_p, C.c_int]),
    "egotap_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "egotap_set_pu_chain": (C.c_int, [C.c_void_p, C.c_int]),
    "egotap_pu_chain_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "egotap_debug_pu_drop_workgroups": (C.c_int, [C.c_void_p, C.c_int]),
    "egotap_debug_gemm_bk": (C.c_int, [C.c_int]),
    "egotap_debug_conv_addressing": (C.c_int, [C.c_int]),
    "egotap_set_weight_scratch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "egotap_set_act_scratch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "egotap_hm_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]),
    "egotap_hm_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_size_t, C.c_void_p]),
    "egotap_hm_intermediate": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]),
    "egotap_hm_forward_bnbatch_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "egotap_hm_forward_bnbatch_intermediate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]),
    "egotap_hm_forward_bnbatch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                            C.c_size_t, C.c_void_p]),
    "egotap_linear_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p] * 5 + [C.c_int, C.c_void_p]),
    "egotap_gemm_tile_name": (C.c_char_p, [C.c_int]),
    "egotap_linear_bf16_dma": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]),
    "egotap_layernorm_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "egotap_attention_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "egotap_pose_metrics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "egotap_pose_metrics_batch_axes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "egotap_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "egotap_synth_heatmaps": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "egotap_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "egotap_timing_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "egotap_timing_detail": (C.c_char_p, [C.c_void_p]),
    # ---- training-step operators
    "egotap_train_gemm_nt": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "egotap_debug_wgrad_splits": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_int)]),
    "egotap_train_gemm_tn_bias": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_train_gemm_tn": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_train_colsum": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_train_transpose": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "egotap_train_add_inplace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "egotap_train_patch_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6),
    "egotap_train_patch_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "egotap_train_tokens_scatter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "egotap_train_layernorm_fwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_float, C.c_void_p]),
    "egotap_train_layernorm_bwd": (C.c_int, [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_train_bn_lrelu_fwd": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_train_bn_lrelu_bwd": (C.c_int, [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_train_qkv_fwd": (C.c_int, [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_void_p]),
    "egotap_train_attention_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "egotap_train_attention_bwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "egotap_train_pu_saved_bytes": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "egotap_train_pu_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_train_pu_bwd_ws_bytes": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]),
    "egotap_train_pu_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_train_pose_head_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "egotap_train_pose_head_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p]),
    "egotap_train_pose_loss": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "egotap_train_adamw": (C.c_int, [C.c_void_p] * 4 + [C.c_int64] + [C.c_double] * 5 + [C.c_int, C.c_void_p]),
    # ---- bf16-storage operators
    "egotap_bf16_gemm_nt": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "egotap_bf16_gemm_tn": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_bf16_layernorm_fwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_float, C.c_void_p]),
    "egotap_bf16_layernorm_bwd": (C.c_int, [C.c_void_p] * 11 + [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_bf16_colsum": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_bf16_prep_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "egotap_bf16_from_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "egotap_bf16_attention_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "egotap_bf16_attention_bwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "egotap_bf16_attention_bwd_bias": (C.c_int, [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_bf16_fc1_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "egotap_bf16_patch_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "egotap_bf16_fc1_wgrad": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_bf16_fc1_dgrad_tokens": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "egotap_train_adamw_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_double] * 5 + [C.c_int, C.c_void_p]),
    "egotap_bind_grad": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "egotap_lift_train_bytes": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "egotap_lift_forward_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_lift_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_int, C.c_void_p]),
    # ---- heatmap-estimator training operators
    "egotap_hmtrain_conv_fwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 7 + [C.c_int64] * 3 + [C.c_void_p]),
    "egotap_hm_conv_bn_fwd": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 7 + [C.c_int64] * 3 + [C.c_void_p]),
    "egotap_hm_stem_bn_fwd": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_void_p]),
    "egotap_hmtrain_set_pack_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "egotap_hmtrain_pack_bytes": (C.c_size_t, []),
    "egotap_hmtrain_stem_fwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p]),
    "egotap_hmtrain_bn2d_fwd": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 3 + [C.c_int64] * 3 + [C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_hmtrain_bn2d_bwd": (C.c_int, [C.c_void_p] * 10 + [C.c_int] * 3 + [C.c_int64] * 2 + [C.c_int] * 3 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_hmtrain_chansum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_hmtrain_conv_wt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "egotap_hmtrain_zero_upsample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "egotap_hmtrain_conv_wgrad": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_int64] * 2 + [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egotap_hmtrain_relu_bwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_int64] * 3 + [C.c_void_p]),
    "egotap_hmtrain_maxpool_bwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_void_p]),
    "egotap_hmtrain_upsample_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "egotap_hmtrain_maxpool_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "egotap_hmtrain_upsample_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "egotap_hmtrain_mse": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
}


def exported_symbols():
    """names declared in include/egotap.h that the library must export"""
    return sorted(_PROTOS)


def load(build_if_missing: bool = True):
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime: import it first so this process has exactly one libamdhip64
    # (loading ours before torch's gives two runtimes and "no ROCm-capable device" at the first launch)
    import torch  # noqa: F401
    path = _build.LIB
    if build_if_missing and not os.path.exists(path):
        # several ranks of one node may get here together (torch.distributed.run): one builds, the others wait on the lock
        import fcntl
        with open(path + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                if not os.path.exists(path):
                    path = _build.build()
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)
    if not os.path.exists(path):
        raise EgotapError(f"{path} is missing: build it with `python -m egotap_amd.build` (hipcc, gfx950)")
    lib = C.CDLL(path)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.egotap_abi_version() != 2:
        raise EgotapError("libegotap_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().egotap_last_error().decode("utf-8", "replace")
        raise EgotapError(f"{ERR_NAMES.get(rc, rc)}: {msg}")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ----------------------------------------------------------------------------------------- single operators
def _need_cuda_f32(*ts):
    import torch
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise EgotapError("egotap_amd ops need contiguous float32 tensors on the GPU (no CPU fallback)")


def linear(x, w, b, epi: str = "bias", residual=None, bn=None, tile: int = 0):
    """y = epi(x @ w.T + b) on the fp32 MFMA GEMM.  epi: bias | residual | gelu | bn_lrelu."""
    import torch
    code = {"bias": 0, "residual": 1, "gelu": 2, "bn_lrelu": 3}[epi]
    M, K = x.shape
    N = w.shape[0]
    g = beta = mean = var = None
    if bn is not None:
        g, beta, mean, var = bn
    _need_cuda_f32(x, w, b, residual, g, beta, mean, var)
    y = torch.empty((M, N), device=x.device, dtype=torch.float32)
    check(load().egotap_linear_f32(_ptr(x), _ptr(w), _ptr(b), _ptr(y), M, N, K, code, _ptr(residual), _ptr(g), _ptr(beta),
                                   _ptr(mean), _ptr(var), tile, _stream()))
    return y


def linear_bf16_dma(x_bf16, w_bf16, b):
    """y = x @ w.T + b on the LDS-DMA bf16 kernel; x [M,K], w [N,K] torch.bfloat16 (caller-owned copies), b fp32 -> fp32"""
    import torch
    if not (x_bf16.is_cuda and w_bf16.is_cuda and x_bf16.dtype == torch.bfloat16 and w_bf16.dtype == torch.bfloat16
            and x_bf16.is_contiguous() and w_bf16.is_contiguous()):
        raise EgotapError("linear_bf16_dma needs contiguous bfloat16 operands on the GPU")
    _need_cuda_f32(b)
    (M, K), N = x_bf16.shape, w_bf16.shape[0]
    y = torch.empty((M, N), device=x_bf16.device, dtype=torch.float32)
    check(load().egotap_linear_bf16_dma(_ptr(x_bf16), _ptr(w_bf16), _ptr(b), _ptr(y), M, N, K, _stream()))
    return y


def pose_metrics(pred, gt, want_aligned: bool = False, reference_batch_axes: bool = False):
    """per-sample (mpjpe [B], pa_mpjpe [B][, aligned [B,J,3]]) of poses [B,J,3] in the input units (egotap_pose_metrics).
    reference_batch_axes: for a batch of 2 or 3 frames return what utils/util.py:328-379 returns there -- line 337 skips its transpose
    and aligns the wrong axes (egotap_pose_metrics_batch_axes); other batch sizes are unaffected, as in the reference."""
    import torch
    pred, gt = pred.detach().float().contiguous(), gt.detach().float().contiguous()
    _need_cuda_f32(pred, gt)
    if pred.shape != gt.shape or pred.dim() != 3 or pred.shape[2] != 3:
        raise ValueError(f"expected pred, gt [B, J, 3]; got {tuple(pred.shape)}, {tuple(gt.shape)}")
    B, J = pred.shape[0], pred.shape[1]
    e, pa = torch.empty(B, device=pred.device), torch.empty(B, device=pred.device)
    al = torch.empty_like(pred) if want_aligned else None
    fn = load().egotap_pose_metrics_batch_axes if reference_batch_axes and B in (2, 3) else load().egotap_pose_metrics
    check(fn(_ptr(pred), _ptr(gt), B, J, _ptr(e), _ptr(pa), _ptr(al), _stream()))
    return (e, pa, al) if want_aligned else (e, pa)


KINEMATIC_PARENTS = {          # utils/util.py:51-52
    "UnrealEgo": [0, 0, 1, 1, 2, 3, 4, 5, 2, 3, 8, 9, 10, 11, 12, 13],
    "EgoCap": [0, 0, 1, 2, 3, 4, 1, 6, 7, 8, 2, 10, 11, 12, 6, 14, 15, 16],
}


def synth_heatmaps(pts2d_left, pts2d_right, pose3d, joint_preset: str = "UnrealEgo", res: int = 64):
    """Ground-truth heatmaps on the device (egotap_synth_heatmaps): joints [B, J+1, 2] x 2 eyes in the 1024-pixel frame and
    gt_local_pose [B, J+1, 3] -> dict with the data loader's keys plus ``cat`` [B, 6J, res, res], the lifting head's input."""
    import torch
    parents = KINEMATIC_PARENTS[joint_preset]
    J1 = len(parents)
    J = J1 - 1
    pl, pr, p3 = (t.detach().float().contiguous() for t in (pts2d_left, pts2d_right, pose3d))
    _need_cuda_f32(pl, pr, p3)
    B = pl.shape[0]
    if tuple(pl.shape) != (B, J1, 2) or tuple(pr.shape) != (B, J1, 2) or tuple(p3.shape) != (B, J1, 3):
        raise ValueError(f"expected [B, {J1}, 2] x 2 and [B, {J1}, 3] for {joint_preset}")
    dev = pl.device
    par = torch.tensor(parents, dtype=torch.int32, device=dev)
    cat = torch.empty((B, 6 * J, res, res), device=dev)
    plen = torch.empty((B, 2, J), device=dev)
    theta = torch.empty((B, J), device=dev)
    check(load().egotap_synth_heatmaps(_ptr(pl), _ptr(pr), _ptr(p3), _ptr(par), B, J, res, _ptr(cat), _ptr(plen), _ptr(theta), _stream()))
    return {"cat": cat, "gt_heatmap_left": cat[:, :J], "gt_heatmap_right": cat[:, J:2 * J],
            "gt_limb_heatmap_left": cat[:, 2 * J:4 * J], "gt_limb_heatmap_right": cat[:, 4 * J:],
            "gt_plength_left": plen[:, 0].repeat(1, 2), "gt_plength_right": plen[:, 1].repeat(1, 2), "gt_limb_theta": theta}


def layernorm(x, gamma, beta, eps: float = 1e-12):
    import torch
    _need_cuda_f32(x, gamma, beta)
    y = torch.empty_like(x)
    rows = x.numel() // x.shape[-1]
    check(load().egotap_layernorm_f32(_ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), rows, x.shape[-1], eps, _stream()))
    return y


def attention(qkv, B: int, N: int, heads: int, precision: str = "f32"):
    """qkv [B*N, 3*heads*128] (q|k|v) -> ctx [B*N, heads*128]"""
    import torch
    _need_cuda_f32(qkv)
    ctx = torch.empty((B * N, heads * 128), device=qkv.device, dtype=torch.float32)
    check(load().egotap_attention(_ptr(qkv), _ptr(ctx), B, N, heads, PRECISIONS[precision], _stream()))
    return ctx
