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


def gemm_nt(x, w, bias, epi="bf16", aux=None, out=None, out1=None, colsum_out=None):
    """OUT = epi(x[M,K] @ w[N,K]^T) on gemm_bf16s_kernel.  x, w bf16; epi: bf16 | residual (aux f32 R) | gelu_save (returns z, h) |
    gelu_grad (aux bf16 z; colsum_out fp32 [N]: also the column sums of the stored output, i.e. the bias gradient of the layer whose
    output gradient this is -- they leave the GEMM's epilogue as per-wave partial sums, finished by a small fp32 column sum) | f32"""
    _bf(x, "x"); _bf(w, "w")
    M, K = x.shape
    N = w.shape[0]
    dev = x.device
    f32_out = epi in ("residual", "f32")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32 if f32_out else torch.bfloat16, device=dev)
    if epi == "gelu_save" and out1 is None:
        out1 = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    if colsum_out is not None:
        if epi != "gelu_grad":
            raise ValueError("colsum_out rides on the gelu_grad epilogue")
        out1 = torch.empty((2 * ((M + 255) // 256), N), dtype=torch.float32, device=dev)      # one row of partial sums per 128-row wave block
    _lib.check(_lib.load().egotap_bf16_gemm_nt(_p(x), x.stride(0), _p(w), _p(bias), M, N, K, EPI[epi], _p(aux), _p(out), _p(out1),
                                               out.stride(0), _s()))
    if colsum_out is not None:
        from .train_ops import colsum as _colsum_f32
        _colsum_f32(out1, colsum_out, out1.shape[0], N)
    return (out, out1) if epi == "gelu_save" else out


_zero_pages = {}


def zero_page(device):
    """a small all-zero device buffer (the TN kernel fetches rows past M from it)"""
    key = str(device)
    if key not in _zero_pages:
        _zero_pages[key] = torch.zeros(4096, dtype=torch.uint8, device=device)
    return _zero_pages[key]


def gemm_tn(dy, x, dw, accumulate=False):
    """dw[N,K] (+)= dy[M,N]^T @ x[M,K]; dy, x bf16, dw fp32"""
    from .train_ops import _scratch
    _bf(dy, "dy"); _bf(x, "x")
    M, N = dy.shape
    K = x.shape[1]
    ws = _scratch.get(max(64 << 20, 4 * N * K * 8) + 8192, dy.device)      # + the check-in counters behind the slabs
    _lib.check(_lib.load().egotap_bf16_gemm_tn(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(dw), M, N, K, int(accumulate), _p(zero_page(dy.device)),
                                               _p(ws), ws.numel(), _s()))
    return dw


def _ws(nbytes, device):
    from .train_ops import _scratch
    return _scratch.get(nbytes, device)


def layernorm_fwd(x, g, b, eps=1e-12, want_stats=True):
    """x fp32 [rows, 1024] -> (y bf16, mean, rstd)"""
    rows = x.shape[0]
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty_like(mean) if want_stats else None
    _lib.check(_lib.load().egotap_bf16_layernorm_fwd(_p(x), _p(y), _p(g), _p(b), _p(mean), _p(rstd), rows, eps, _s()))
    return y, mean, rstd


def layernorm_bwd(x, dy, g, mean, rstd, dgamma, dbeta, dres=None, dcolsum=None, want_bf16=True, accumulate=False):
    """returns (dx fp32, dx bf16 or None); dcolsum (+)= column sums of dx"""
    rows = x.shape[0]
    dx = torch.empty_like(x)
    dxb = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    nb = (rows + 63) // 64
    ws = _ws((3 * nb + 3 + 3 * ((nb + 63) // 64)) * 4096 + 4096, x.device)
    _lib.check(_lib.load().egotap_bf16_layernorm_bwd(_p(x), _p(dy), _p(g), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dxb), _p(dgamma), _p(dbeta),
                                                     _p(dcolsum), rows, int(accumulate), _p(ws), ws.numel(), _s()))
    return dx, dxb


def colsum(y, out, accumulate=False):
    M, N = y.shape
    ws = _ws(64 << 20, y.device)
    _lib.check(_lib.load().egotap_bf16_colsum(_p(y), y.stride(0), _p(out), M, N, int(accumulate), _p(ws), ws.numel(), _s()))
    return out


def prep_weight(w, wb, wt=None):
    """w fp32 [N, K] -> wb bf16 [N, K] (and wt bf16 [K, N], which may be a column slice of a wider matrix)"""
    N, K = w.shape[0], w.numel() // w.shape[0]
    _lib.check(_lib.load().egotap_bf16_prep_weight(_p(w), _p(wb), _p(wt), N, K, wt.stride(0) if wt is not None else N, _s()))


def from_f32(src, dst=None):
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    _lib.check(_lib.load().egotap_bf16_from_f32(_p(src), _p(dst), src.numel(), _s()))
    return dst


def attention_fwd(qkv, B, N, heads, want_lse=True):
    ctx = torch.empty((B * N, heads * 128), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty(B * heads * N, dtype=torch.float32, device=qkv.device) if want_lse else None
    _lib.check(_lib.load().egotap_bf16_attention_fwd(_p(qkv), _p(ctx), _p(lse), B, N, heads, _s()))
    return ctx, lse


def attention_bwd(qkv, ctx, dctx, lse, B, N, heads, bias_grads=None):
    """dqkv; bias_grads = (dq_bias, dk_bias, dv_bias) fp32 [heads * 128] each: also the column sums of dqkv, from the kernels' epilogues"""
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    if bias_grads is None:
        _lib.check(_lib.load().egotap_bf16_attention_bwd(_p(qkv), _p(ctx), _p(dctx), _p(lse), _p(delta), _p(dqkv), B, N, heads, _s()))
        return dqkv
    ws = _ws(B * (N // 32) * 3 * heads * 128 * 4 + (64 << 20), qkv.device)
    _lib.check(_lib.load().egotap_bf16_attention_bwd_bias(_p(qkv), _p(ctx), _p(dctx), _p(lse), _p(delta), _p(dqkv), _p(bias_grads[0]), _p(bias_grads[1]),
                                                          _p(bias_grads[2]), B, N, heads, _p(ws), ws.numel(), _s()))
    return dqkv


def fc1_fwd(h, which, src, w, bias, B, T):
    z = torch.empty((B * T, 2048), dtype=torch.float32, device=src.device)
    _lib.check(_lib.load().egotap_bf16_fc1_fwd(h, which, _p(src), _p(w), _p(bias), _p(z), B, _s()))
    return z


def fc1_wgrad(h, which, dz, src, dw, B):
    ws = _ws(max(64 << 20, 4 * dw.numel() * 2), dz.device)
    _lib.check(_lib.load().egotap_bf16_fc1_wgrad(h, which, _p(dz), _p(src), _p(dw), B, _p(zero_page(dz.device)), _p(ws), ws.numel(), _s()))
    return dw


def fc1_dgrad_tokens(h, dz, wt, B, seq, D):
    dtok = torch.empty((B * seq, D), dtype=torch.bfloat16, device=dz.device)
    _lib.check(_lib.load().egotap_bf16_fc1_dgrad_tokens(h, _p(dz), _p(wt), _p(dtok), B, _s()))
    return dtok
