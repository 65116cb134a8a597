"""GPU parity of the heatmap estimator (HIP conv kernels through the C ABI) against the golden vectors captured
from the reference's HeatMap_UnrealEgo_Shared (over our ResNet-18 stand-in) and against the float64 oracle."""
import os

import numpy as np
import pytest
import torch

from egotap_amd.synthetic import synth_input

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rgb(name, B):
    return torch.from_numpy(synth_input(name, (B, 3, 256, 256), -2.0, 2.0))


@pytest.mark.parametrize("which,n_hm", [("pos", 15), ("rot", 30)])
def test_hm_forward_matches_golden(which, n_hm):
    from gpu_util import hm_net
    g = np.load(os.path.join(GOLD, f"hm_full_{which}.npz"))
    net, _ = hm_net(which)
    left, right = _rgb("rgb_left", 1).cuda(), _rgb("rgb_right", 1).cuda()
    y = net(left, right)
    torch.cuda.synchronize()
    assert tuple(y.shape) == tuple(g["out_shape"])
    yc = y.cpu()
    np.testing.assert_allclose(yc[0, 0].numpy(), g["out_ch0"], atol=3e-4, rtol=1e-4)
    np.testing.assert_allclose(yc[0, -1].numpy(), g["out_last"], atol=3e-4, rtol=1e-4)
    np.testing.assert_allclose(yc.reshape(-1)[::97].numpy(), g["out_sample"], atol=3e-4, rtol=1e-4)
    for k in ("conv_up3", "conv_up2", "conv_up1"):
        got = net.intermediate(k, 1).cpu().reshape(-1)[::997].numpy()
        np.testing.assert_allclose(got, g[k + "_sample"], atol=3e-4, rtol=1e-4, err_msg=k)
    # backbone pyramid of the RIGHT eye (image n = 2*0 + 1)
    for i, c in enumerate((64, 64, 128, 256, 512)):
        t = net.intermediate(f"layer{i}", 1).cpu()
        right_eye = t.reshape(2, -1)[1]
        np.testing.assert_allclose(right_eye[::997].numpy(), g[f"pyr{i}_sample"], atol=2e-4, rtol=1e-4, err_msg=f"layer{i}")


@pytest.mark.parametrize("which,B", [("pos", 3), ("rot", 2)])
def test_hm_forward_matches_oracle(which, B):
    from gpu_util import hm_net
    from oracle import hm_ref as H
    net, sd_np = hm_net(which)
    left, right = _rgb(f"rgbL_{which}_{B}", B), _rgb(f"rgbR_{which}_{B}", B)
    sd = H.to_torch_sd(sd_np, torch.float64)
    with torch.no_grad():
        ref = H.hm_forward(left.double(), right.double(), sd)
    y = net(left.cuda(), right.cuda())
    torch.cuda.synchronize()
    err = (y.cpu().double() - ref).abs().max().item()
    assert err < 1e-4 * max(1.0, ref.abs().max().item()), f"max err {err:.3e} (max |ref| {ref.abs().max().item():.3e})"


def test_hm_writes_into_channel_slice_and_batch_independent():
    from gpu_util import hm_net
    net, _ = hm_net("pos")
    left, right = _rgb("rgb_left", 1).cuda(), _rgb("rgb_right", 1).cuda()
    alone = net(left, right).clone()
    cat = torch.full((5, 90, 64, 64), 7.0, device="cuda")
    net.forward_into(left.repeat(5, 1, 1, 1), right.repeat(5, 1, 1, 1), cat, channel_offset=0)
    torch.cuda.synchronize()
    assert torch.equal(cat[:, :30], alone.expand(5, -1, -1, -1))      # bit-identical per sample, any batch
    assert float((cat[:, 30:] - 7.0).abs().max()) == 0.0              # nothing outside the slice is touched


def test_hm_forward_512_rgb_egocap_matches_oracle():
    """BASELINE config 5 geometry: EgoCap preset, 512x512 RGB -> 128x128 heatmaps (128-wide conv instantiations,
    stride-2 convs from 128 to 64)."""
    from gpu_util import hm_net
    from oracle import hm_ref as H
    net, sd_np = hm_net("pos", preset="EgoCap", hm=128)
    left = torch.from_numpy(synth_input("rgbL_ec512", (1, 3, 512, 512), -2.0, 2.0))
    right = torch.from_numpy(synth_input("rgbR_ec512", (1, 3, 512, 512), -2.0, 2.0))
    sd = H.to_torch_sd(sd_np, torch.float64)
    with torch.no_grad():
        ref = H.hm_forward(left.double(), right.double(), sd)
    y = net(left.cuda(), right.cuda())
    torch.cuda.synchronize()
    assert tuple(y.shape) == (1, 34, 128, 128)
    err = (y.cpu().double() - ref).abs().max().item()
    assert err < 1e-4 * max(1.0, ref.abs().max().item()), f"max err {err:.3e} (max |ref| {ref.abs().max().item():.3e})"


@pytest.mark.parametrize("which,B", [("pos", 3), ("rot", 1)])
def test_hm_forward_bf16x3_mode_matches_oracle(which, B):
    """opt-in mode: 3x3 stride-1 convs with >= 128 output channels on the bf16 matrix cores as hi+lo splits (channels-last LDS
    image, repacked weights, Cin = 1540 tail slab); same gate as fp32.  Switching back restores the exact result bit for bit."""
    from gpu_util import hm_net
    from oracle import hm_ref as H
    net, sd_np = hm_net(which)
    left, right = _rgb(f"rgbL_{which}_{B}", B), _rgb(f"rgbR_{which}_{B}", B)
    sd = H.to_torch_sd(sd_np, torch.float64)
    with torch.no_grad():
        ref = H.hm_forward(left.double(), right.double(), sd)
    exact = net(left.cuda(), right.cuda()).clone()
    try:
        net.set_precision("bf16x3")
        fast = net(left.cuda(), right.cuda()).clone()
        again = net(left.cuda(), right.cuda()).clone()
    finally:
        net.set_precision("f32")
    back = net(left.cuda(), right.cuda()).clone()
    torch.cuda.synchronize()
    err = (fast.cpu().double() - ref).abs().max().item()
    assert err < 1e-4 * max(1.0, ref.abs().max().item()), f"max err {err:.3e} (max |ref| {ref.abs().max().item():.3e})"
    assert torch.equal(fast, again) and torch.equal(back, exact) and not torch.equal(fast, exact)


@pytest.mark.parametrize("which,preset,hm,B", [("pos", "UnrealEgo", 64, 3), ("rot", "UnrealEgo", 64, 2), ("rot", "EgoCap", 128, 1)])
def test_hm_forward_bf16_channels_last_decoder_against_float64_oracle(which, preset, hm, B):
    """EGOTAP_PREC_BF16: the decoder runs on bf16 channels-last activations, every convolution as an implicit GEMM on the bf16-storage
    GEMM kernel (conv_bf16s.h: 3x3 taps through a loader with a zero page for the padding, 1x1 lateral convs with padded N, the
    concat buffers' channel slices written in place, conv_heatmap back to fp32 NCHW).  Against the FLOAT64 ORACLE: relative L2 below
    2 % (about ten bf16 roundings deep), every frame of the batch (ragged last tile at the 8x8 level), bit-reproducible, and a frame's
    result does not depend on the batch it is in."""
    from gpu_util import hm_net
    from oracle import hm_ref as H
    net, sd_np = hm_net(which, preset=preset, hm=hm)
    S = 4 * hm
    left = torch.from_numpy(synth_input(f"rgbL_cl_{which}{hm}", (B, 3, S, S), -2.0, 2.0))
    right = torch.from_numpy(synth_input(f"rgbR_cl_{which}{hm}", (B, 3, S, S), -2.0, 2.0))
    with torch.no_grad():
        ref = H.hm_forward(left.double(), right.double(), H.to_torch_sd(sd_np, torch.float64))
    try:
        net.set_precision("bf16")
        low = net(left.cuda(), right.cuda())
        again = net(left.cuda(), right.cuda())
        last = net(left[B - 1:].cuda(), right[B - 1:].cuda())
    finally:
        net.set_precision("f32")
    assert torch.equal(low, again) and torch.equal(low[B - 1:], last)
    low = low.double().cpu()
    assert tuple(low.shape) == tuple(ref.shape)
    rels = [float((low[b] - ref[b]).norm() / ref[b].norm()) for b in range(B)]
    print(f"bf16 estimator ({which}, {preset}, {hm}): relative L2 against the float64 oracle per frame {['%.2e' % r for r in rels]}")
    for b, rel in enumerate(rels):
        assert 1e-5 < rel < 2e-2, (b, rel)


def test_hm_forward_bf16_mode_against_float64_oracle():
    """plain bf16 operands (2^-9 per rounding) in the 3x3 convolutions of the estimator, checked against the FLOAT64 ORACLE: the
    heatmaps stay within 3 % relative L2 of it (fp32 mode: 1e-6), and are not the fp32 result"""
    from gpu_util import hm_net
    from oracle import hm_ref as H
    net, sd_np = hm_net("pos")
    left, right = _rgb("rgb_left", 1), _rgb("rgb_right", 1)
    with torch.no_grad():
        ref = H.hm_forward(left.double(), right.double(), H.to_torch_sd(sd_np, torch.float64))
    exact = net(left.cuda(), right.cuda()).double().cpu()
    try:
        net.set_precision("bf16")
        low = net(left.cuda(), right.cuda()).double().cpu()
    finally:
        net.set_precision("f32")
    rel = float((low - ref).norm() / ref.norm())
    assert float((exact - ref).norm() / ref.norm()) < 1e-5
    assert 1e-5 < rel < 3e-2, rel
