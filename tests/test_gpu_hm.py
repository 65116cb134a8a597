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


@pytest.mark.parametrize("which,B,model_name", [("pos", 3, "resnet18"), ("rot", 2, "resnet18"), ("rot", 3, "resnet34"),
                                                ("pos", 2, "resnet50"), ("rot", 1, "resnet101")])
def test_hm_forward_matches_oracle(which, B, model_name):
    """resnet34 (--model_name, net_architecture.py:59-60): BasicBlocks (3, 4, 6, 3) per stage, same decoder (feature_scale 1);
    resnet50 / resnet101 (:61-64, 108-111): Bottleneck blocks, every decoder width x 4 (conv_up3: 6160 -> 4096 channels) -- the forward is
    composed from the operator entry points (egotap_hm_conv_bn_fwd ...), fp32, eval mode"""
    from gpu_util import hm_net
    from oracle import hm_ref as H
    net, sd_np = hm_net(which, model_name=model_name)
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
    # [r4] a few frames split the input channels of the small-map convolutions over the chip (conv_f32.h: the number of ranges follows the
    # batch, so the fp32 summation order does too): there a sample's heatmaps agree to rounding, run to run bit for bit ...
    scale = float(alone.abs().max())
    assert float((cat[:, :30] - alone.expand(5, -1, -1, -1)).abs().max()) < 1e-5 * scale
    assert torch.equal(cat[0, :30], cat[4, :30]) and torch.equal(alone, net(left, right))
    assert float((cat[:, 30:] - 7.0).abs().max()) == 0.0              # nothing outside the slice is touched
    # ... and from 65 frames on no convolution is split (the last one to fill half the chip is layer4's 1 x 1 lateral convolution: four 8 x 8 images per
    # pixel tile): bit-identical per sample, any batch
    a72 = net(left.repeat(72, 1, 1, 1), right.repeat(72, 1, 1, 1))
    a80 = net(left.repeat(80, 1, 1, 1), right.repeat(80, 1, 1, 1))
    assert torch.equal(a72[0], a80[79]) and torch.equal(a72[71], a80[0])
    assert float((a72[:1] - alone).abs().max()) < 1e-5 * scale


@pytest.mark.parametrize("model_name", ["resnet18", "resnet50"])
def test_hm_forward_512_rgb_egocap_matches_oracle(model_name):
    """BASELINE config 5 geometry: EgoCap preset, 512x512 RGB -> 128x128 heatmaps (128-wide conv instantiations,
    stride-2 convs from 128 to 64); [r3] also through the Bottleneck backbone's operator composition (1x1 convolutions on 128 x 128 maps)."""
    from gpu_util import hm_net
    from oracle import hm_ref as H
    net, sd_np = hm_net("pos", preset="EgoCap", hm=128, model_name=model_name)
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


@pytest.mark.parametrize("which,preset,hm,B,model_name", [("pos", "UnrealEgo", 64, 3, "resnet18"), ("rot", "UnrealEgo", 64, 2, "resnet18"),
                                                          ("rot", "EgoCap", 128, 1, "resnet18"), ("pos", "UnrealEgo", 64, 2, "resnet34")])
def test_hm_forward_bf16_channels_last_decoder_against_float64_oracle(which, preset, hm, B, model_name):
    """EGOTAP_PREC_BF16: the decoder runs on bf16 channels-last activations, every convolution as an implicit GEMM on the bf16-storage
    GEMM kernel (conv_bf16s.h: 3x3 taps through a loader with a zero page for the padding, 1x1 lateral convs with padded N, the
    concat buffers' channel slices written in place, conv_heatmap back to fp32 NCHW).  Against the FLOAT64 ORACLE: relative L2 below
    2 % (about ten bf16 roundings deep), every frame of the batch (ragged last tile at the 8x8 level), bit-reproducible run to run; a frame's
    result moves with the batch size only by flipped bf16 roundings."""
    from gpu_util import hm_net
    from oracle import hm_ref as H
    net, sd_np = hm_net(which, preset=preset, hm=hm, model_name=model_name)
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
    assert torch.equal(low, again)
    # [r4] which kernel a convolution runs on follows its pixel count (few pixels: 32-deep K-tiles split over the chip; many: 64-deep), so a frame's
    # bf16 heatmaps move with the batch size by flipped bf16 roundings (the fp32 mode is the one that is batch-independent to 1e-5): a wrong
    # routing would be O(1)
    rel_batch = float((low[B - 1:].double() - last.double()).norm() / last.double().norm())
    assert rel_batch < 1e-2, rel_batch
    # ... and EVERY routing is gated against the float64 oracle by itself (round-4 advice: the 1e-2 between batch sizes is half the whole bf16
    # budget, a mis-routed small layer could hide in it): the same frame alone in its batch -- the serving route: K split over the chip in
    # layer3 / layer4 and the three decoder convolutions (hm_dec_ksplit, fixed-order reduce) -- must meet the oracle gate of the batched route
    rel_alone = float((last[0].double().cpu() - ref[B - 1]).norm() / ref[B - 1].norm())
    assert 1e-5 < rel_alone < (2e-2 if model_name == "resnet18" else 3e-2), rel_alone
    low = low.double().cpu()
    assert tuple(low.shape) == tuple(ref.shape)
    rels = [float((low[b] - ref[b]).norm() / ref[b].norm()) for b in range(B)]
    print(f"bf16 estimator ({which}, {preset}, {hm}, {model_name}): relative L2 against the float64 oracle per frame {['%.2e' % r for r in rels]}")
    for b, rel in enumerate(rels):
        assert 1e-5 < rel < (2e-2 if model_name == "resnet18" else 3e-2), (b, rel)         # resnet34: twice the backbone depth


def test_hm_forward_bf16_large_batch_routes_layers_2_to_4_through_the_64_deep_gemm():
    """[r4] At serving / training batches the BasicBlock 3x3 convolutions of layer2 (256 x 128 tile), layer3 and layer4 run on the 64-deep
    GEMM with the X64ConvE loader (stride 1 and 2, 64-channel weight slabs); below ~130 frames layer4 still takes the 32-deep kernel's
    split-K path, so the small-batch tests never reach it.  B = 136: against the same forward pinned to the 32-deep kernels
    (egotap_debug_gemm_bk(32): another summation order, so bf16 roundings flip -- relative L2 of a few 1e-3, a wrong tap / slab / eye
    would be O(1)), run-to-run bits, and the last frame against the float64 oracle.  [r5] The default addressing of the convolution operands
    (X64ConvES / X64Conv3S: one wave-uniform origin + 32-bit lane offsets) against the per-lane pointer form: the same bits."""
    from gpu_util import hm_net
    from oracle import hm_ref as H
    from egotap_amd import lib
    L = lib.load()
    net, sd_np = hm_net("rot")
    B = 136
    left = torch.from_numpy(synth_input("rgbL_big", (8, 3, 256, 256), -2.0, 2.0)).repeat(B // 8, 1, 1, 1)
    right = torch.from_numpy(synth_input("rgbR_big", (8, 3, 256, 256), -2.0, 2.0)).repeat(B // 8, 1, 1, 1)
    left[B - 1] = torch.from_numpy(synth_input("rgbL_big_last", (3, 256, 256), -2.0, 2.0))
    with torch.no_grad():
        ref = H.hm_forward(left[B - 1:].double(), right[B - 1:].double(), H.to_torch_sd(sd_np, torch.float64))[0]
    lc, rc = left.cuda(), right.cuda()
    try:
        net.set_precision("bf16")
        lib.check(L.egotap_debug_gemm_bk(32))
        old = net(lc, rc)
        lib.check(L.egotap_debug_gemm_bk(0))
        new = net(lc, rc)
        again = net(lc, rc)
        lib.check(L.egotap_debug_conv_addressing(1))       # [r5] per-lane pointers instead of a scalar origin + lane offsets: the same bytes fetched
        by_pointer = net(lc, rc)
    finally:
        lib.check(L.egotap_debug_conv_addressing(0))
        lib.check(L.egotap_debug_gemm_bk(0))
        net.set_precision("f32")
    assert torch.equal(new, again)
    assert torch.equal(new, by_pointer)
    assert not torch.equal(new, old)                        # the two routings are different kernels
    new, old = new.double().cpu(), old.double().cpu()
    rel_route = float((new - old).norm() / old.norm())
    rel_last = float((new[B - 1] - ref).norm() / ref.norm())
    rel_first = float((new[0] - new[8]).norm())             # frames 0 and 8 are the same image pair: the same bits wherever they sit in the batch
    print(f"64-deep vs 32-deep routing: relative L2 {rel_route:.2e}; last frame against float64: {rel_last:.2e}")
    assert rel_route < 1e-2 and 1e-5 < rel_last < 2e-2 and rel_first == 0.0, (rel_route, rel_last, rel_first)


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


@pytest.mark.parametrize("preset,hm,B", [("UnrealEgo", 64, 3), ("EgoCap", 128, 2)])
def test_bf16_fused_stem_and_maxpool_kernel(preset, hm, B):
    """csrc/stem_bf16s.h: conv 7x7/2 + BatchNorm(eval) + ReLU + MaxPool(3, 2, 1) as one bf16-MFMA kernel (net_architecture.py:69-70),
    read back from the workspace (bf16 [B * hm * hm, 2 x 64], eye-interleaved) against float64 arithmetic on the SAME bf16-rounded
    inputs and weights, rounded to bf16 once at the same place: equal up to one bf16 ulp where the fp32 accumulation order decides a
    rounding; image borders (max-pool padding, zero halo), both eyes, the segment seams and, at 512 x 512, four segments per row."""
    import ctypes as C
    from gpu_util import hm_net
    from egotap_amd import lib
    L = lib.load()
    net, sd_np = hm_net("pos", preset=preset, hm=hm)
    S = 4 * hm
    left = torch.from_numpy(synth_input(f"rgbL_stem_{hm}", (B, 3, S, S), -2.0, 2.0))
    right = torch.from_numpy(synth_input(f"rgbR_stem_{hm}", (B, 3, S, S), -2.0, 2.0))
    try:
        net.set_precision("bf16")
        fused = net(left.cuda(), right.cuda())
        off, n = C.c_size_t(), C.c_int64()
        lib.check(L.egotap_hm_intermediate(net._ensure_handle(), B, b"pool0", C.byref(off), C.byref(n)))
        got = net._ws[off.value: off.value + 2 * n.value].view(torch.bfloat16).reshape(B, hm, hm, 2, 64).double().cpu()
        assert torch.isfinite(fused).all()
    finally:
        net.set_precision("f32")
    rb = lambda t: t.float().bfloat16().double()
    w = rb(torch.from_numpy(sd_np["backbone.backbone.backbone.conv1.weight"]))
    bn = {k: torch.from_numpy(sd_np["backbone.backbone.backbone.bn1." + k]).double() for k in ("weight", "bias", "running_mean", "running_var")}
    sc = (bn["weight"].float() / torch.sqrt(bn["running_var"].float() + 1e-5)).double()          # the kernel folds in fp32
    sh = (bn["bias"].float() - bn["running_mean"].float() * sc.float()).double()
    for eye, img in enumerate((left, right)):
        z = torch.nn.functional.conv2d(rb(img), w, stride=2, padding=3)
        y = rb(torch.relu(z * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)))
        want = torch.nn.functional.max_pool2d(y, 3, 2, 1).permute(0, 2, 3, 1)                 # [B, hm, hm, 64]
        have = got[:, :, :, eye]
        err = (have - want).abs()
        # one bf16 ulp of the value, plus the fp32 accumulation noise of the 147-term sum where BatchNorm's shift cancels it
        tol = 2.0 ** -7 * want.abs() + 4e-6 * float(z.abs().max() * sc.abs().max())
        bad = err > tol
        i = int(err.argmax())
        assert not bool(bad.any()), (eye, int(bad.sum()), float(err.reshape(-1)[i]), float(want.reshape(-1)[i]), float(have.reshape(-1)[i]))
        assert float((err == 0).double().mean()) > 0.98, float((err == 0).double().mean())


@pytest.mark.parametrize("preset,hm,B", [("UnrealEgo", 64, 3), ("EgoCap", 128, 1), ("UnrealEgo", 64, 40)])
def test_bf16_direct_conv64_kernel_against_float64_on_the_same_operands(preset, hm, B):
    """csrc/conv64_bf16s.h (layer1's four 3x3 convolutions 64 -> 64 with BatchNorm, residual and ReLU: halo tile in LDS, weights in
    registers): layer1's output read back from the workspace against float64 arithmetic on the SAME operands -- the pooled bf16 map
    the GPU produced, weights rounded to bf16, the BatchNorm fold in fp32, every block output rounded to bf16 once where the kernel
    stores it -- image borders and tile seams included; equal up to bf16 roundings the fp32 summation order decides (two BasicBlocks
    deep: a flipped rounding of one convolution moves the next one's input); bit-reproducible.
    B = 40: 1280 tiles on 256 persistent workgroups (five tiles each: the halo double buffer, the counted wait that leaves the previous
    tile's stores in flight) and, in the stem in front, 640 row-group runs on 512 workgroups.  (Round 3 compared against the
    implicit-GEMM kernel this one replaced; that switch left the library in round 4.)"""
    import ctypes as C
    from gpu_util import hm_net
    from egotap_amd import lib
    L = lib.load()
    net, sd_np = hm_net("rot", preset=preset, hm=hm)
    S = 4 * hm
    nb = min(B, 8)          # distinct frames (the hash generator is slow): larger batches cycle through them with a per-frame scale
    left = torch.from_numpy(synth_input(f"rgbL_c64_{hm}", (nb, 3, S, S), -2.0, 2.0)).cuda()
    right = torch.from_numpy(synth_input(f"rgbR_c64_{hm}", (nb, 3, S, S), -2.0, 2.0)).cuda()
    if B > nb:
        idx = torch.arange(B, device="cuda") % nb
        gain = (1.0 + 0.01 * torch.arange(B, device="cuda", dtype=torch.float32)).view(B, 1, 1, 1)
        left, right = (left[idx] * gain).contiguous(), (right[idx] * gain).contiguous()

    def inter(name):
        off, n = C.c_size_t(), C.c_int64()
        lib.check(L.egotap_hm_intermediate(net._ensure_handle(), B, name, C.byref(off), C.byref(n)))
        return net._ws[off.value: off.value + 2 * n.value].view(torch.bfloat16).clone()
    try:
        net.set_precision("bf16")
        y_new = net(left, right)
        l_new, p0 = inter(b"layer1_bf16"), inter(b"pool0")
        y_again = net(left, right)
        assert torch.equal(l_new, inter(b"layer1_bf16")) and torch.equal(y_new, y_again)
    finally:
        net.set_precision("f32")
    rb = lambda t: t.float().bfloat16().double()
    # [B * hm * hm, 2 x 64] eye-interleaved -> [2B, 64, hm, hm] (image n = 2 b + eye); float64 on the GPU (80 images of 64 x 64 x 64)
    x = p0.reshape(B, hm, hm, 2, 64).permute(0, 3, 4, 1, 2).reshape(2 * B, 64, hm, hm).double()
    pre = "backbone.backbone.backbone.layer1."
    for blk in (0, 1):
        idt = x
        for cv in (1, 2):
            w = rb(torch.from_numpy(sd_np[f"{pre}{blk}.conv{cv}.weight"])).cuda()
            bn = {k: torch.from_numpy(sd_np[f"{pre}{blk}.bn{cv}.{k}"]) for k in ("weight", "bias", "running_mean", "running_var")}
            sc = bn["weight"] / torch.sqrt(bn["running_var"] + 1e-5)                       # the fold runs in fp32 on the device
            sh = bn["bias"] - bn["running_mean"] * sc
            z = torch.nn.functional.conv2d(x, w, padding=1) * sc.double().cuda().view(1, -1, 1, 1) + sh.double().cuda().view(1, -1, 1, 1)
            if cv == 2:
                z = z + idt
            x = rb(torch.relu(z))
    want = x.reshape(B, 2, 64, hm, hm).permute(0, 3, 4, 1, 2).reshape(-1, 128)
    have = l_new.reshape(-1, 128).double()
    err = (have - want).abs()
    scale = float(want.abs().max())
    exact = float((err == 0).double().mean())
    one_ulp = float((err <= 2.0 ** -7 * want.abs() + 1e-30).double().mean())
    print(f"direct conv64 vs float64 ({preset}, B = {B}): equal on {exact * 100:.2f} % of the elements, within one bf16 ulp on {one_ulp * 100:.3f} %, "
          f"max |diff| {float(err.max()):.3e} (max |value| {scale:.2f})")
    assert float(err.max()) <= 2.0 ** -5 * scale, (float(err.max()), scale)             # a few ulps of the largest values at worst
    assert exact > 0.95 and one_ulp > 0.995, (exact, one_ulp)
    rel = float((have - want).norm() / want.norm())
    assert rel < 2e-3, rel


def test_bf16_fused_stem_many_runs_per_workgroup():
    """the fused stem at 40 stereo frames: 640 (image, row group) runs on 512 persistent workgroups, so some workgroups walk two runs
    (patch restaging, carry buffers and the pooling ring reused across runs).  EVERY image of both eyes against float64 on the same
    bf16 operands (round 3 checked three images that way and the rest against the two-kernel form, which left the library in round 4)."""
    import ctypes as C
    from gpu_util import hm_net
    from egotap_amd import lib
    L = lib.load()
    net, sd_np = hm_net("pos")
    B, hm, S = 40, 64, 256
    base_l = torch.from_numpy(synth_input("rgbL_stem_many", (4, 3, S, S), -2.0, 2.0)).cuda()
    base_r = torch.from_numpy(synth_input("rgbR_stem_many", (4, 3, S, S), -2.0, 2.0)).cuda()
    idx = torch.arange(B, device="cuda") % 4
    gain = (1.0 + 0.02 * torch.arange(B, device="cuda", dtype=torch.float32)).view(B, 1, 1, 1)
    left, right = (base_l[idx] * gain).contiguous(), (base_r[idx] * gain).contiguous()
    try:
        net.set_precision("bf16")
        net(left, right)
        off, n = C.c_size_t(), C.c_int64()
        lib.check(L.egotap_hm_intermediate(net._ensure_handle(), B, b"pool0", C.byref(off), C.byref(n)))
        fused = net._ws[off.value: off.value + 2 * n.value].view(torch.bfloat16).reshape(B, hm, hm, 2, 64).double()
    finally:
        net.set_precision("f32")
    rb = lambda t: t.float().bfloat16().double()
    w = rb(torch.from_numpy(sd_np["backbone.backbone.backbone.conv1.weight"])).cuda()
    bn = {k: torch.from_numpy(sd_np["backbone.backbone.backbone.bn1." + k]) for k in ("weight", "bias", "running_mean", "running_var")}
    sc = (bn["weight"] / torch.sqrt(bn["running_var"] + 1e-5))
    sh = (bn["bias"] - bn["running_mean"] * sc).double().cuda()
    sc = sc.double().cuda()
    for eye, img in enumerate((left, right)):
        for lo in range(0, B, 8):                                        # float64 on the GPU, eight images at a time
            z = torch.nn.functional.conv2d(rb(img[lo:lo + 8]), w, stride=2, padding=3)
            y = rb(torch.relu(z * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)))
            want = torch.nn.functional.max_pool2d(y, 3, 2, 1).permute(0, 2, 3, 1)
            err = (fused[lo:lo + 8, :, :, eye] - want).abs()
            tol = 2.0 ** -7 * want.abs() + 4e-6 * float(z.abs().max() * sc.abs().max())
            assert not bool((err > tol).any()), (lo, eye, float(err.max()))
