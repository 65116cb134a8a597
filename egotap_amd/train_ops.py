"""Thin Python wrappers over the C-ABI training operators (include/egotap.h, "training-step operators").

Tensors are allocated with torch (plumbing); all arithmetic happens in libegotap_hip.so.  Used by
egotap_amd/training.py (the autograd glue) and by the per-operator parity tests.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as _lib

LD_PLAIN, LD_PATCH, LD_TOKENS, LD_ROT, LD_STEREO, LD_STEREO_GATED = range(6)
TE_NONE, TE_BIAS, TE_BIAS_RES, TE_BIAS_GELU_SAVE, TE_ACCUM, TE_GELU_GRAD = range(6)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Scratch:
    """Grow-only device scratch for split-M slabs / column-sum partials (never shrinks, reused by every call)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes: int, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        return self.buf


_scratch = Scratch()


def gemm_nt(h, x, w, b, M, N, K, loader=LD_PLAIN, epi=TE_BIAS, r=None, z=None, aux=None, lda=0, Bsz=0, out=None):
    """y[M,N] = epi(A(x)[M,K] @ w[N,K]^T (+ b))"""
    y = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().egotap_train_gemm_nt(h, loader, _p(x), lda, _p(aux), _p(w), _p(b), _p(y), M, N, K, epi, _p(r), _p(z), Bsz, _s()))
    return y


def gemm_tn(h, dy, x, dw, M, N, K, loader=LD_PLAIN, accumulate=False, aux=None, ldy=0, Bsz=0):
    """dw[N,K] (+)= dy[M,N]^T @ A(x)[M,K]"""
    ws = _scratch.get(max(64 << 20, 4 * N * K * 8), dy.device)
    _lib.check(_lib.load().egotap_train_gemm_tn(h, loader, _p(dy), ldy, _p(x), _p(aux), _p(dw), M, N, K, int(accumulate), Bsz, _p(ws),
                                                ws.numel(), _s()))
    return dw


def gemm_tn_bias(h, dy, x, dw, db, M, N, K, accumulate=False, ldy=0):
    """dw[N,K] (+)= dy[M,N]^T @ x[M,K] and db[N] (+)= column sums of dy: weight and bias gradient of a Linear layer in one call"""
    ws = _scratch.get(max(64 << 20, 4 * N * (K + 1) * 20), dy.device)       # room for 20 partial slabs: never what limits the split count
    _lib.check(_lib.load().egotap_train_gemm_tn_bias(h, _p(dy), ldy, _p(x), _p(dw), _p(db), M, N, K, int(accumulate), _p(ws), ws.numel(), _s()))
    return dw, db


def colsum(y, out, M, N, accumulate=False, ldy=0):
    ws = _scratch.get(64 << 20, y.device)
    _lib.check(_lib.load().egotap_train_colsum(_p(y), ldy, _p(out), M, N, int(accumulate), _p(ws), ws.numel(), _s()))
    return out


def transpose(w, out=None, ldo=0):
    """out[C, ldo>=R] = w[R,C]^T"""
    R, Cc = w.shape
    if out is None:
        out = torch.empty((Cc, R), dtype=torch.float32, device=w.device)
    _lib.check(_lib.load().egotap_train_transpose(_p(w), _p(out), R, Cc, ldo, _s()))
    return out


def add_inplace(out, x):
    _lib.check(_lib.load().egotap_train_add_inplace(_p(out), _p(x), out.numel(), _s()))
    return out


def layernorm_fwd(x, g, b, eps=1e-12):
    rows = x.shape[0]
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    _lib.check(_lib.load().egotap_train_layernorm_fwd(_p(x), _p(y), _p(g), _p(b), _p(mean), _p(rstd), rows, eps, _s()))
    return y, mean, rstd


def layernorm_bwd(x, dy, g, mean, rstd, dgamma, dbeta, dres=None, accumulate=False):
    rows = x.shape[0]
    dx = torch.empty_like(x)
    nb = (rows + 63) // 64
    ws = _scratch.get((nb + 1 + (nb + 63) // 64) * 2048 * 4 + 4096, x.device)
    _lib.check(_lib.load().egotap_train_layernorm_bwd(_p(x), _p(dy), _p(g), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dgamma), _p(dbeta),
                                                      rows, int(accumulate), _p(ws), ws.numel(), _s()))
    return dx


def bn_lrelu_fwd(z, gamma, beta, run_mean, run_var, eps=1e-5, momentum=0.1):
    R, Cc = z.shape
    y = torch.empty_like(z)
    mean = torch.empty(Cc, dtype=torch.float32, device=z.device)
    rstd = torch.empty_like(mean)
    ws = _scratch.get(((R + 255) // 256 * 2 * Cc + 2 * Cc) * 4 + 4096, z.device)
    _lib.check(_lib.load().egotap_train_bn_lrelu_fwd(_p(z), _p(y), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(run_mean), _p(run_var), R, Cc,
                                                     eps, momentum, _p(ws), ws.numel(), _s()))
    return y, mean, rstd


def bn_lrelu_bwd(z, y, dy, gamma, mean, rstd, dgamma, dbeta, accumulate=False):
    R, Cc = z.shape
    dz = torch.empty_like(z)
    ws = _scratch.get(((R + 255) // 256 * 2 * Cc + 2 * Cc) * 4 + 4096, z.device)
    _lib.check(_lib.load().egotap_train_bn_lrelu_bwd(_p(z), _p(y), _p(dy), _p(gamma), _p(mean), _p(rstd), _p(dz), _p(dgamma), _p(dbeta), R, Cc,
                                                     int(accumulate), _p(ws), ws.numel(), _s()))
    return dz


def attention_fwd(qkv, B, N, heads, precision="f32"):
    ctx = torch.empty((B * N, heads * 128), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((B * heads * N,), dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.load().egotap_train_attention_fwd(_p(qkv), _p(ctx), _p(lse), B, N, heads, _lib.PRECISIONS[precision], _s()))
    return ctx, lse


def attention_bwd(qkv, ctx, dctx, lse, B, N, heads, precision="f32"):
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    _lib.check(_lib.load().egotap_train_attention_bwd(_p(qkv), _p(ctx), _p(dctx), _p(lse), _p(delta), _p(dqkv), B, N, heads, _lib.PRECISIONS[precision], _s()))
    return dqkv


def pose_loss(h, pred, gt, lambda_mpjpe=0.1, lambda_cos_sim=-0.01):
    """returns (losses[2] = (loss_pose, loss_cos_sim), dpred [2, B, J, 3] = (d loss_pose / d pred, d loss_cos_sim / d pred))"""
    B = pred.shape[0]
    dpred = torch.empty((2,) + tuple(pred.shape), dtype=torch.float32, device=pred.device)
    out = torch.empty(2, dtype=torch.float32, device=pred.device)
    partial = torch.empty((B, 2), dtype=torch.float32, device=pred.device)
    _lib.check(_lib.load().egotap_train_pose_loss(h, _p(pred), _p(gt), _p(dpred), _p(out), _p(partial), B, lambda_mpjpe, lambda_cos_sim, _s()))
    return out, dpred


def adamw(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01):
    _lib.check(_lib.load().egotap_train_adamw(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step, _s()))
