"""Thin Python wrappers of the heatmap-estimator training operators (include/egotap.h egotap_hmtrain_*).  Tensors are NCHW
fp32 on the GPU; ``view`` arguments are (tensor, channel_offset, channels): a channel slice of a larger [N, Ctot, H, W]
buffer addressed in place through its image stride.  No PyTorch compute: everything lands in libegotap_hip.so."""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as _lib
from .train_ops import Scratch, _p, _s

_scratch = Scratch()


def _ws(dev, nbytes=256 << 20):
    t = _scratch.get(nbytes, dev)
    return C.c_void_p(t.data_ptr()), t.numel()


class View:
    """channel slice [c0, c0 + C) of a contiguous [N, Ctot, H, W] tensor"""

    def __init__(self, t, c0=0, C_=None):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 4
        self.t, self.c0, self.C = t, c0, (t.shape[1] - c0 if C_ is None else C_)
        self.N, self.H, self.W = t.shape[0], t.shape[2], t.shape[3]
        self.istride = t.shape[1] * self.H * self.W
        assert 0 <= c0 and c0 + self.C <= t.shape[1]

    @property
    def ptr(self):
        return C.c_void_p(self.t.data_ptr() + 4 * self.c0 * self.H * self.W)

    def tensor(self):
        return self.t[:, self.c0:self.c0 + self.C]


def V(t, c0=0, C_=None):
    return t if isinstance(t, View) else View(t, c0, C_)


_zeros = {}


def zeros_vec(n, dev):
    k = (str(dev),)
    if k not in _zeros or _zeros[k].numel() < n:
        _zeros[k] = torch.zeros(max(n, 4096), device=dev)
    return _zeros[k]


def conv_fwd(h, x, w, y, bias=None, res=None, taps=9, stride=1, relu=False):
    """y = [relu](conv(x, w) + bias [+ res]); x, y, res Views; w [Cout, Cin, k, k]"""
    x, y = V(x), V(y)
    res = V(res) if res is not None else None
    Cout, Cin = w.shape[0], w.shape[1]
    assert x.C == Cin and y.C == Cout and y.N == x.N and y.W * stride == x.W
    b = bias if bias is not None else zeros_vec(Cout, x.t.device)
    _lib.check(_lib.load().egotap_hmtrain_conv_fwd(h, x.ptr, _p(w), _p(b), res.ptr if res is not None else None, y.ptr, x.N, Cin, Cout, y.W, taps,
                                                   stride, int(relu), x.istride, y.istride, res.istride if res is not None else 0, _s()))


def conv_bn_fwd(h, x, w, bn, y, res=None, taps=9, stride=1, relu=True):
    """y = [relu](BatchNorm_eval(conv(x, w)) [+ res]); bn = (gamma, beta, running_mean, running_var); x, y, res Views"""
    x, y = V(x), V(y)
    res = V(res) if res is not None else None
    Cout, Cin = w.shape[0], w.shape[1]
    assert x.C == Cin and y.C == Cout and y.N == x.N and y.W * stride == x.W
    _lib.check(_lib.load().egotap_hm_conv_bn_fwd(h, x.ptr, _p(w), _p(bn[0]), _p(bn[1]), _p(bn[2]), _p(bn[3]), res.ptr if res is not None else None, y.ptr,
                                                 x.N, Cin, Cout, y.W, taps, stride, int(relu), x.istride, y.istride,
                                                 res.istride if res is not None else 0, _s()))


def stem_bn_fwd(left, right, w, bn, y):
    """y [2B, 64, S0/2, S0/2] = relu(BatchNorm_eval(conv7x7/2(image n = 2b + eye)))"""
    _lib.check(_lib.load().egotap_hm_stem_bn_fwd(_p(left), _p(right), _p(w), _p(bn[0]), _p(bn[1]), _p(bn[2]), _p(bn[3]), _p(y), left.shape[0],
                                                 left.shape[2], _s()))


def conv_dgrad(h, dy, w, dx, taps=9, stride=1, accumulate=False):
    """dx (+)= d/dx of conv(x, w): the forward kernels on flipped / channel-swapped weights (stride 2: dY spread over the even pixels)"""
    dy, dx = V(dy), V(dx)
    Cout, Cin = w.shape[0], w.shape[1]
    if (Cout * taps) % 4 != 0:          # the forward kernels stage weights by float4: pad the contracted channels (30 -> 32) with zeros
        pad = (-Cout) % 4
        w = torch.cat([w, torch.zeros((pad,) + tuple(w.shape[1:]), device=w.device)], 0).contiguous()
        dyp = torch.zeros((dy.N, Cout + pad, dy.H, dy.W), device=w.device)
        dyp[:, :Cout] = dy.tensor()
        dy, Cout = V(dyp), Cout + pad
    wt = torch.empty((Cin, Cout) + tuple(w.shape[2:]), device=w.device)
    _lib.check(_lib.load().egotap_hmtrain_conv_wt(_p(w), _p(wt), Cout, Cin, taps, _s()))
    src = dy
    if stride == 2:
        up = torch.empty((dy.N, Cout, 2 * dy.H, 2 * dy.W), device=w.device)
        _lib.check(_lib.load().egotap_hmtrain_zero_upsample(dy.ptr, _p(up), dy.N, Cout, dy.H, dy.istride, up.shape[1] * up.shape[2] * up.shape[3], _s()))
        src = V(up)
    conv_fwd(h, src, wt, dx, res=dx if accumulate else None, taps=taps, stride=1)


def conv_wgrad(dy, x, dw, ks=3, stride=1, accumulate=False, precision="f32"):
    dy, x = V(dy), V(x)
    Cout, Cin = dw.shape[0], dw.shape[1]
    assert dy.C == Cout and x.C == Cin
    ws, n = _ws(dw.device)
    _lib.check(_lib.load().egotap_hmtrain_conv_wgrad(dy.ptr, x.ptr, _p(dw), x.N, Cin, Cout, dy.W, ks, stride, dy.istride, x.istride, int(accumulate),
                                                     _lib.PRECISIONS[precision], ws, n, _s()))


def bn2d_fwd(z, y, gamma, beta, run_mean, run_var, res=None, relu=True, eps=1e-5, momentum=0.1):
    z, y = V(z), V(y)
    res = V(res) if res is not None else None
    mean, rstd = torch.empty(z.C, device=z.t.device), torch.empty(z.C, device=z.t.device)
    ws, n = _ws(z.t.device)
    _lib.check(_lib.load().egotap_hmtrain_bn2d_fwd(z.ptr, y.ptr, res.ptr if res is not None else None, _p(gamma), _p(beta), _p(mean), _p(rstd),
                                                   _p(run_mean), _p(run_var), z.N, z.C, z.H * z.W, z.istride, y.istride,
                                                   res.istride if res is not None else 0, int(relu), eps, momentum, ws, n, _s()))
    return mean, rstd


def bn2d_bwd(z, y, dy, gamma, mean, rstd, dz, dgamma, dbeta, dres=None, relu=True, accumulate=False, dres_accumulate=False):
    z, dy, dz = V(z), V(dy), V(dz)
    assert dz.istride == z.istride and (dres is None or V(dres).istride == z.istride)
    ws, n = _ws(z.t.device)
    _lib.check(_lib.load().egotap_hmtrain_bn2d_bwd(z.ptr, V(y).ptr if y is not None else None, dy.ptr, _p(gamma), _p(mean), _p(rstd), dz.ptr,
                                                   V(dres).ptr if dres is not None else None, _p(dgamma), _p(dbeta), z.N, z.C, z.H * z.W, z.istride,
                                                   dy.istride, int(relu), int(accumulate), int(dres_accumulate), ws, n, _s()))


def chansum(dy, out, accumulate=False):
    dy = V(dy)
    ws, n = _ws(out.device)
    _lib.check(_lib.load().egotap_hmtrain_chansum(dy.ptr, _p(out), dy.N, dy.C, dy.H * dy.W, dy.istride, int(accumulate), ws, n, _s()))


def relu_bwd(y, dy, dz):
    y, dy, dz = V(y), V(dy), V(dz)
    _lib.check(_lib.load().egotap_hmtrain_relu_bwd(y.ptr, dy.ptr, dz.ptr, y.N, y.C, y.H * y.W, y.istride, dy.istride, dz.istride, _s()))


def maxpool_bwd(x, dy, dx):
    _lib.check(_lib.load().egotap_hmtrain_maxpool_bwd(_p(x), _p(dy), _p(dx), x.shape[0] * x.shape[1], x.shape[2], _s()))


def upsample_bwd(dy, dx):
    dy, dx = V(dy), V(dx)
    _lib.check(_lib.load().egotap_hmtrain_upsample_bwd(dy.ptr, dx.ptr, dx.N, dx.C, dx.H, dy.istride, dx.istride, _s()))


def maxpool_fwd(x, y):
    _lib.check(_lib.load().egotap_hmtrain_maxpool_fwd(_p(x), _p(y), x.shape[0] * x.shape[1], x.shape[2], _s()))


def upsample_fwd(x, y):
    x, y = V(x), V(y)
    _lib.check(_lib.load().egotap_hmtrain_upsample_fwd(x.ptr, y.ptr, x.N, x.C, x.H, x.istride, y.istride, _s()))


def stem_fwd(left, right, w, z):
    _lib.check(_lib.load().egotap_hmtrain_stem_fwd(_p(left), _p(right), _p(w), _p(z), left.shape[0], left.shape[2], _s()))


def mse(pred, gt, plen, lam):
    """returns (loss [1], dpred) for lam * (mean_left + mean_right)((pred - gt)^2 / plen)"""
    B, Cn, H, W = pred.shape
    dpred, loss = torch.empty_like(pred), torch.empty(1, device=pred.device)
    ws, n = _ws(pred.device)
    _lib.check(_lib.load().egotap_hmtrain_mse(_p(pred), _p(gt), _p(plen) if plen is not None else None, _p(dpred), _p(loss), B, Cn, H * W, float(lam),
                                              ws, n, _s()))
    return loss, dpred


def mse_halves(pred, gt, plen, lam):
    """(loss_left, loss_right) [2] and dpred for lam * MSE(left half) + lam * MSE(right half) of pred / gt [B, 2n, H, W]
    (the reference keeps the two terms as separate loss_* attributes); plen [B, 2n] or None"""
    B, C2, Hh, W = pred.shape
    n = C2 // 2
    dpred = torch.empty_like(pred)
    losses = torch.empty(2, device=pred.device)
    ws, nb = _ws(pred.device)
    for half in range(2):
        p_ = pred[:, half * n:(half + 1) * n].contiguous()
        g_ = gt[:, half * n:(half + 1) * n].contiguous()
        l_ = plen[:, half * n:(half + 1) * n].contiguous() if plen is not None else None
        d_ = torch.empty_like(p_)
        # one half = "left + right" of a [B, n] problem whose second half is empty: run it as Cn = n with the mean over n channels
        _lib.check(_lib.load().egotap_hmtrain_mse(_p(p_), _p(g_), _p(l_) if l_ is not None else None, _p(d_), _p(losses[half:half + 1]), B, n, Hh * W,
                                                  float(lam) * 0.5, ws, nb, _s()))
        dpred[:, half * n:(half + 1) * n] = d_
    return losses, dpred
