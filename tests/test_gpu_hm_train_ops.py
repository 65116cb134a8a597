"""GPU parity of the heatmap-estimator training operators (C ABI egotap_hmtrain_*) against float64 torch autograd on the CPU."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g, dtype=torch.float64) * (hi - lo) + lo).float()


def _close(got, ref64, atol, rtol=1e-4, msg=""):
    got = got.detach().cpu().double()
    err = (got - ref64).abs()
    tol = atol + rtol * ref64.abs()
    assert bool((err <= tol).all()), f"{msg} max err {err.max().item():.3e} (max ref {ref64.abs().max().item():.3e})"


def _handle():
    from gpu_util import hm_net
    net, _ = hm_net("pos")
    return net._ensure_handle()


CONVS = [  # (ks, stride, Cin, Cout, Wout, N)
    (3, 1, 64, 64, 64, 2), (3, 1, 128, 128, 32, 3), (3, 1, 100, 200, 16, 4), (3, 1, 512, 512, 8, 2),
    (3, 2, 64, 128, 32, 2), (3, 2, 128, 256, 16, 3), (3, 2, 256, 512, 8, 2),
    (1, 1, 128, 128, 64, 2), (1, 1, 256, 260, 32, 2), (1, 1, 512, 516, 16, 3), (1, 1, 1024, 1024, 8, 2),
    (1, 2, 64, 128, 32, 2), (1, 2, 128, 256, 16, 2), (1, 2, 256, 512, 8, 2),
    # [r3] the 2 x 2 wave layout of the layers with at most 64 output channels (ragged on both sides), odd image counts for the row splits,
    # the stem
    (3, 1, 70, 50, 64, 3), (1, 1, 512, 30, 64, 3), (1, 1, 100, 64, 64, 1), (3, 1, 640, 512, 64, 1), (7, 2, 3, 64, 128, 3),
]


@pytest.mark.parametrize("ks,stride,Cin,Cout,W,N", CONVS)
def test_conv_wgrad_and_dgrad(ks, stride, Cin, Cout, W, N):
    """weight gradient (implicit GEMM over pixels, split over images) and input gradient (forward kernels on flipped weights,
    stride 2 through zero-upsampled dY) against autograd of F.conv2d; ragged channel counts; accumulate mode"""
    from egotap_amd import hm_ops as H
    h = _handle()
    x, w = _rand((N, Cin, W * stride, W * stride), 1), _rand((Cout, Cin, ks, ks), 2, -0.1, 0.1)
    dy = _rand((N, Cout, W, W), 3)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    F.conv2d(xr, wr, None, stride, (ks - 1) // 2).backward(dy.double())
    dw = torch.full((Cout, Cin, ks, ks), 3.0, device="cuda")
    H.conv_wgrad(dy.cuda(), x.cuda(), dw, ks=ks, stride=stride)
    scale = float(wr.grad.abs().mean())
    _close(dw, wr.grad, atol=2e-4 * scale + 1e-5, msg="dw")
    first = dw.clone()
    H.conv_wgrad(dy.cuda(), x.cuda(), dw, ks=ks, stride=stride, accumulate=True)
    _close(dw, 2 * wr.grad, atol=4e-4 * scale + 2e-5, msg="dw accumulate")
    again = torch.empty_like(dw)
    H.conv_wgrad(dy.cuda(), x.cuda(), again, ks=ks, stride=stride)
    assert torch.equal(first, again)
    if ks == 7:
        return                                     # the stem has no input gradient (the images are data)
    dx = torch.full((N, Cin, W * stride, W * stride), 7.0, device="cuda")
    H.conv_dgrad(h, dy.cuda(), w.cuda(), dx, taps=ks * ks, stride=stride)
    _close(dx, xr.grad, atol=2e-4 * float(xr.grad.abs().mean()) + 1e-5, msg="dx")


def test_stem_raw_and_wgrad():
    from egotap_amd import hm_ops as H
    B, S0 = 2, 256
    l, r, w = _rand((B, 3, S0, S0), 1), _rand((B, 3, S0, S0), 2), _rand((64, 3, 7, 7), 3, -0.1, 0.1)
    x = torch.stack([l, r], 1).reshape(2 * B, 3, S0, S0)
    z = torch.empty((2 * B, 64, S0 // 2, S0 // 2), device="cuda")
    H.stem_fwd(l.cuda(), r.cuda(), w.cuda(), z)
    wr = w.double().requires_grad_(True)
    ref = F.conv2d(x.double(), wr, None, 2, 3)
    _close(z, ref.detach(), 2e-5)
    dy = _rand(tuple(z.shape), 4)
    ref.backward(dy.double())
    dw = torch.empty((64, 3, 7, 7), device="cuda")
    H.conv_wgrad(dy.cuda(), x.cuda(), dw, ks=7, stride=2)
    _close(dw, wr.grad, atol=2e-4 * float(wr.grad.abs().mean()), msg="stem dw")


@pytest.mark.parametrize("N,C,Hs,relu,with_res", [(4, 64, 32, True, False), (6, 128, 16, True, True), (2, 516, 8, False, False), (3, 64, 64, True, True)])
def test_bn2d_fwd_bwd(N, C, Hs, relu, with_res):
    from egotap_amd import hm_ops as H
    z, res = _rand((N, C, Hs, Hs), 1, -2, 2), _rand((N, C, Hs, Hs), 2)
    g, b = _rand((C,), 3, 0.5, 1.5), _rand((C,), 4)
    rm, rv = _rand((C,), 5), _rand((C,), 6, 0.5, 2.0)
    dy = _rand((N, C, Hs, Hs), 7)
    zr, gr, br, rr = (t.double().requires_grad_(True) for t in (z, g, b, res))
    rm64, rv64 = rm.double().clone(), rv.double().clone()
    y = F.batch_norm(zr, rm64, rv64, gr, br, True, 0.1, 1e-5)
    if with_res:
        y = y + rr
    if relu:
        y = F.relu(y)
    y.backward(dy.double())
    yd = torch.empty_like(z, device="cuda")
    rmd, rvd = rm.cuda(), rv.cuda()
    mean, rstd = H.bn2d_fwd(z.cuda(), yd, g.cuda(), b.cuda(), rmd, rvd, res=res.cuda() if with_res else None, relu=relu)
    _close(yd, y.detach(), 2e-5)
    _close(rmd, rm64, 1e-6)
    _close(rvd, rv64, 1e-5)
    dz, dres = torch.empty_like(yd), (torch.empty_like(yd) if with_res else None)
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    H.bn2d_bwd(z.cuda(), yd, dy.cuda(), g.cuda(), mean, rstd, dz, dg, db, dres=dres, relu=relu)
    _close(dz, zr.grad, 2e-5, rtol=1e-3)
    _close(dg, gr.grad, 2e-3, rtol=1e-4)
    _close(db, br.grad, 2e-3, rtol=1e-4)
    if with_res:
        _close(dres, rr.grad, 1e-6)


def test_pointwise_backward_ops():
    from egotap_amd import hm_ops as H
    # max-pool 3/2/1 (ties broken like torch: first maximum in the window scan)
    x = _rand((2, 8, 32, 32), 1)
    x[0, 0, 4:8, 4:8] = 0.5                       # plateau: ties
    xr = x.double().requires_grad_(True)
    dy = _rand((2, 8, 16, 16), 2)
    F.max_pool2d(xr, 3, 2, 1).backward(dy.double())
    dx = torch.empty_like(x, device="cuda")
    H.maxpool_bwd(x.cuda(), dy.cuda(), dx)
    _close(dx, xr.grad, 1e-6)
    # [r3] strip kernel: a last strip that is not full (24 rows = 16 + 8), the training size (128) and one plane row per strip element
    for side, seed in ((24, 5), (128, 6), (8, 7)):
        x = _rand((3, 5, side, side), seed)
        x[1, 2, 2:7, 1:6] = 0.25
        xr = x.double().requires_grad_(True)
        dy = _rand((3, 5, side // 2, side // 2), seed + 10)
        F.max_pool2d(xr, 3, 2, 1).backward(dy.double())
        dx = torch.full_like(x, 9.0, device="cuda")
        H.maxpool_bwd(x.cuda(), dy.cuda(), dx)
        torch.cuda.synchronize()
        _close(dx, xr.grad, 1e-6, msg=f"maxpool backward, side {side}")
    # bilinear x2 upsample, align_corners=True, into / out of channel slices of wider buffers
    src = _rand((3, 6, 8, 8), 3).double().requires_grad_(True)
    dup = _rand((3, 10, 16, 16), 4)
    F.interpolate(src, scale_factor=2, mode="bilinear", align_corners=True).backward(dup[:, 2:8].double())
    dsrc = torch.zeros((3, 9, 8, 8), device="cuda")
    H.upsample_bwd(H.View(dup.cuda(), 2, 6), H.View(dsrc, 1, 6))
    _close(dsrc[:, 1:7], src.grad, 1e-5)
    assert float(dsrc[:, 0].abs().max()) == 0 and float(dsrc[:, 7:].abs().max()) == 0
    # ReLU mask and per-channel sums on slices
    y, d = _rand((2, 12, 16, 16), 5), _rand((2, 12, 16, 16), 6)
    dz = torch.zeros((2, 20, 16, 16), device="cuda")
    H.relu_bwd(H.View(y.cuda(), 4, 8), H.View(d.cuda(), 4, 8), H.View(dz, 10, 8))
    _close(dz[:, 10:18], (d[:, 4:12] * (y[:, 4:12] > 0)).double(), 0)
    out = torch.full((8,), 2.0, device="cuda")
    H.chansum(H.View(d.cuda(), 4, 8), out)
    _close(out, d[:, 4:12].double().sum((0, 2, 3)), 1e-4)
    H.chansum(H.View(d.cuda(), 4, 8), out, accumulate=True)
    _close(out, 2 * d[:, 4:12].double().sum((0, 2, 3)), 2e-4)


@pytest.mark.parametrize("limb", [False, True])
def test_mse_loss(limb):
    """heatmap_shared_model.py:109-151: lambda * (MSE(left) + MSE(right)), limb maps divided by sqrt(gt_plength) first"""
    from egotap_amd import hm_ops as H
    B, Cn, S = 3, 30 if not limb else 60, 64
    pred, gt = _rand((B, Cn, S, S), 1), _rand((B, Cn, S, S), 2)
    plen = _rand((B, Cn), 3, 1.0, 40.0) if limb else None
    pr = pred.double().requires_grad_(True)
    lam = 10.0
    if limb:
        sq = torch.sqrt(plen.double())[..., None, None]
        loss = lam * (F.mse_loss(pr[:, :Cn // 2] / sq[:, :Cn // 2], gt.double()[:, :Cn // 2] / sq[:, :Cn // 2])
                      + F.mse_loss(pr[:, Cn // 2:] / sq[:, Cn // 2:], gt.double()[:, Cn // 2:] / sq[:, Cn // 2:]))
    else:
        loss = lam * (F.mse_loss(pr[:, :Cn // 2], gt.double()[:, :Cn // 2]) + F.mse_loss(pr[:, Cn // 2:], gt.double()[:, Cn // 2:]))
    loss.backward()
    l, dp = H.mse(pred.cuda(), gt.cuda(), plen.cuda() if limb else None, lam)
    np.testing.assert_allclose(float(l), float(loss.detach()), rtol=1e-5)
    _close(dp, pr.grad, 1e-9, rtol=1e-4)


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
@pytest.mark.parametrize("Cin,Cout,W,N", [(64, 64, 64, 2), (128, 128, 32, 3), (100, 200, 16, 4), (640, 512, 64, 1), (1540, 1024, 16, 2)])
def test_conv_wgrad_bf16_modes(mode, Cin, Cout, W, N):
    """3x3 stride-1 weight gradient on the bf16 matrix cores: pre-shifted input rows (shuffled edge pixels, zero halos), ragged
    channel counts (Cin = 100, 1540), image split; bf16x3 against float64 with the 2^-16 error model, bf16 against the float64
    gradient of the rounded operands"""
    from egotap_amd import hm_ops as H
    x, dy = _rand((N, Cin, W, W), 1), _rand((N, Cout, W, W), 3)
    w = torch.zeros((Cout, Cin, 3, 3), dtype=torch.float64, requires_grad=True)
    xs, ds = (x.double(), dy.double()) if mode == "bf16x3" else (x.bfloat16().double(), dy.bfloat16().double())
    F.conv2d(xs, w, None, 1, 1).backward(ds)
    dw = torch.full((Cout, Cin, 3, 3), 3.0, device="cuda")
    H.conv_wgrad(dy.cuda(), x.cuda(), dw, ks=3, stride=1, precision=mode)
    scale = float(w.grad.abs().mean())
    _close(dw, w.grad, atol=(3e-4 if mode == "bf16x3" else 2e-4) * scale + 1e-5, msg="dw " + mode)
    again = torch.empty_like(dw)
    H.conv_wgrad(dy.cuda(), x.cuda(), again, ks=3, stride=1, precision=mode)
    assert torch.equal(dw, again)
