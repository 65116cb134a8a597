"""Python wrappers over the bf16-storage operators of the C ABI (include/egotap.h, "bf16-storage operators"): tensors are allocated
with torch (plumbing), the arithmetic is in libegotap_hip.so.  Used by the bf16 training path (egotap_amd/training.py) and the
operator tests."""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as _lib

EPI = {"bf16": 0, "residual": 1, "gelu_save": 2, "gelu_grad": 3, "f32": 4}


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _bf(t, what):
    if not (t.is_cuda and t.dtype == torch.bfloat16 and t.stride(-1) == 1):
        raise _lib.EgotapError(f"{what}: bfloat16 CUDA tensor with contiguous rows expected")


def gemm_nt(x, w, bias, epi="bf16", aux=None, out=None, out1=None):
    """OUT = epi(x[M,K] @ w[N,K]^T) on gemm_bf16s_kernel.  x, w bf16; epi: bf16 | residual (aux f32 R) | gelu_save (returns z, h) |
    gelu_grad (aux bf16 z) | f32"""
    _bf(x, "x"); _bf(w, "w")
    M, K = x.shape
    N = w.shape[0]
    dev = x.device
    f32_out = epi in ("residual", "f32")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32 if f32_out else torch.bfloat16, device=dev)
    if epi == "gelu_save" and out1 is None:
        out1 = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.load().egotap_bf16_gemm_nt(_p(x), x.stride(0), _p(w), _p(bias), M, N, K, EPI[epi], _p(aux), _p(out), _p(out1),
                                               out.stride(0), _s()))
    return (out, out1) if epi == "gelu_save" else out
