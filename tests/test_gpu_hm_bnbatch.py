"""[r5] egotap_hm_forward_bnbatch: the FROZEN estimators' forward with batch-statistics BatchNorm2d on the bf16 channels-last kernels -- what
the reference computes from RGB while the lifting head trains under --use_amp: train.py:91 model.train() leaves the estimators' BatchNorm2d in
training mode (model/egotap_autoencoder_model.py:127-129 freezes parameters only, :177-216 runs them under autocast), the shared backbone
runs once per eye (model/net_architecture.py:45-50): per-eye statistics, running statistics updated twice per call (left, right).

Gated the way the bf16 training step is (tests/test_gpu_bf16_stages.py): every stage of the backbone against float64 arithmetic on that
stage's OWN inputs as the GPU produced them, rounded to bf16 where the kernels store bf16; then the whole forward against the float64
train-mode oracle, and the wrapper under the reference's flags against the reference wrapper's own fixture."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from egotap_amd.synthetic import synth_hm_state_dict, synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
BB = "backbone.backbone.backbone."


def _rb(t):
    return t.float().bfloat16().double()


def _fresh(which, preset="UnrealEgo", hm=64, model_name="resnet18"):
    """an estimator of its own (the buffers move): hash-RNG weights, bf16 precision"""
    from gpu_util import make_opt
    from egotap_amd import networks
    opt = make_opt(preset, hm)
    if which == "pos":
        opt.num_rot_heatmap = 0
    else:
        opt.num_heatmap = 0
    net = networks.HeatMap_UnrealEgo_Shared(opt, model_name, input_channel_scale=2)
    sd_np = synth_hm_state_dict(net.num_heatmap, f"hm_{which}.", model_name)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    net = net.cuda()
    net.set_precision("bf16")
    return net, sd_np


def _rgb(tag, B, S):
    nb = min(B, 6)          # (the hash generator is slow: larger batches cycle through six frames with a per-frame gain)
    t = torch.from_numpy(synth_input(tag, (nb, 3, S, S), -2.0, 2.0))
    if B > nb:
        idx = torch.arange(B) % nb
        t = (t[idx] * (1.0 + 0.02 * torch.arange(B, dtype=torch.float32)).view(B, 1, 1, 1)).contiguous()
    return t


def _bn_batch(z, g, b):
    """train-mode BatchNorm2d of ONE eye's batch z [B, C, s, s] in float64 -> (normalised, batch mean, biased variance)"""
    mu = z.mean(dim=(0, 2, 3))
    var = z.var(dim=(0, 2, 3), unbiased=False)
    y = (z - mu.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    return y, mu, var


def _running(sd_np, pre, stats, n):
    """running statistics after the left and then the right eye's update (momentum 0.1, unbiased variance)"""
    rm, rv = torch.from_numpy(sd_np[pre + ".running_mean"]).double(), torch.from_numpy(sd_np[pre + ".running_var"]).double()
    for mu, var in stats:
        rm = 0.9 * rm + 0.1 * mu
        rv = 0.9 * rv + 0.1 * var * n / (n - 1)
    return rm, rv


def _eyes(level, B, s, C):
    """bf16 [B * s * s, 2 C] eye-interleaved -> float64 [2][B, C, s, s]"""
    t = level.reshape(B, s, s, 2, C).double()
    return [t[:, :, :, e].permute(0, 3, 1, 2).contiguous() for e in range(2)]


@pytest.mark.parametrize("which,preset,hm,B,model_name", [("pos", "UnrealEgo", 64, 3, "resnet18"), ("rot", "EgoCap", 128, 2, "resnet18"),
                                                          ("rot", "UnrealEgo", 64, 2, "resnet34")])
def test_bnbatch_every_backbone_stage_against_float64_on_its_own_inputs(which, preset, hm, B, model_name):
    """stem (statistics pass + fused BatchNorm / ReLU / max-pool pass) and the four ResNet stages, each against float64 arithmetic on the map the
    GPU fed it: convolution of bf16 operands, z rounded to bf16 where the kernel stores it, per-eye batch statistics of those bf16 values,
    normalise (+ identity) + ReLU, rounded to bf16 -- and every BatchNorm's running_mean / running_var / num_batches_tracked after the call."""
    net, sd_np = _fresh(which, preset, hm, model_name)
    S = 4 * hm
    left, right = _rgb(f"bnb_L_{which}{hm}", B, S), _rgb(f"bnb_R_{which}{hm}", B, S)
    out = torch.empty((B, 2 * net.num_heatmap, hm, hm), device="cuda")
    net.forward_bnbatch_into(left.cuda(), right.cuda(), out, chunk=B)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    bufs = {k: v.detach().double().cpu() for k, v in net.named_buffers()}
    lv = {n: net.bnbatch_intermediate(n, B, B).clone().cpu() for n in ("pool0", "layer1", "layer2", "layer3", "layer4")}
    P = {k: torch.from_numpy(v).double() for k, v in sd_np.items()}

    def check_running(pre, stats, n, tol=2e-3):
        rm, rv = _running(sd_np, pre, stats, n)
        np.testing.assert_allclose(bufs[pre + ".running_mean"].numpy(), rm.numpy(), rtol=tol, atol=tol * float(rm.abs().max()), err_msg=pre)
        np.testing.assert_allclose(bufs[pre + ".running_var"].numpy(), rv.numpy(), rtol=tol, atol=tol * float(rv.abs().max()), err_msg=pre)
        assert int(bufs[pre + ".num_batches_tracked"]) == 2, pre

    # ---- stem: statistics of the fp32 accumulators (the map never reaches HBM), BatchNorm + ReLU + max-pool fused
    w = _rb(P[BB + "conv1.weight"])
    stats, want = [], []
    for img in (left, right):
        z = F.conv2d(_rb(img), w, stride=2, padding=3)
        y, mu, var = _bn_batch(z, P[BB + "bn1.weight"], P[BB + "bn1.bias"])
        stats.append((mu, var))
        want.append(F.max_pool2d(_rb(torch.relu(y)), 3, 2, 1))
    check_running(BB + "bn1", stats, B * (S // 2) ** 2, tol=2e-4)
    have = _eyes(lv["pool0"], B, hm, 64)
    for e in range(2):
        err = (have[e] - want[e]).abs()
        assert float(err.max()) <= 2.0 ** -6 * float(want[e].abs().max()), (e, float(err.max()))
        assert float((err <= 2.0 ** -7 * want[e].abs() + 1e-6).double().mean()) > 0.995
    # ---- stages: each from the GPU's own input map
    stage_ch = (64, 128, 256, 512)
    names = ("pool0", "layer1", "layer2", "layer3", "layer4")
    for i in range(4):
        c, s_in = stage_ch[i], hm >> max(i - 1, 0)
        s_out = hm >> i
        cin = 64 if i == 0 else stage_ch[i - 1]
        x = _eyes(lv[names[i]], B, s_in, cin)
        n = B * s_out * s_out
        for b in range(net.blocks[i]):
            pre = f"{BB}layer{i + 1}.{b}"
            stride = 2 if (b == 0 and i > 0) else 1
            st = {k: [] for k in ("bn1", "bn2", "downsample.1")}
            y = []
            for e in range(2):
                idt = x[e]
                if (pre + ".downsample.0.weight") in P:
                    zd = _rb(F.conv2d(x[e], _rb(P[pre + ".downsample.0.weight"]), None, stride))
                    yd, mu, var = _bn_batch(zd, P[pre + ".downsample.1.weight"], P[pre + ".downsample.1.bias"])
                    st["downsample.1"].append((mu, var))
                    idt = _rb(yd)
                z1 = _rb(F.conv2d(x[e], _rb(P[pre + ".conv1.weight"]), None, stride, 1))
                y1, mu, var = _bn_batch(z1, P[pre + ".bn1.weight"], P[pre + ".bn1.bias"])
                st["bn1"].append((mu, var))
                y1 = _rb(torch.relu(y1))
                z2 = _rb(F.conv2d(y1, _rb(P[pre + ".conv2.weight"]), None, 1, 1))
                y2, mu, var = _bn_batch(z2, P[pre + ".bn2.weight"], P[pre + ".bn2.bias"])
                st["bn2"].append((mu, var))
                y.append(_rb(torch.relu(y2 + idt)))
            for k, v in st.items():
                if v:
                    # the statistics are sums over bf16 values whose roundings the fp32 summation order of the convolution decides: a flipped
                    # rounding upstream moves them at the 1e-3 level (deeper blocks inherit the block before)
                    check_running(f"{pre}.{k}", v, n, tol=1e-2)
            x = y
        have = _eyes(lv[names[i + 1]], B, s_out, c)
        for e in range(2):
            rel = float((have[e] - x[e]).norm() / x[e].norm())
            assert rel < 1e-2, (names[i + 1], e, rel)          # two (resnet34: up to six) blocks deep from the GPU's own input
            print(f"{names[i + 1]} eye {e}: relative L2 against the float64 emulation from the GPU's own input {rel:.2e}")


@pytest.mark.parametrize("which,preset,hm,B", [("pos", "UnrealEgo", 64, 3), ("rot", "UnrealEgo", 64, 2), ("pos", "EgoCap", 128, 2)])
def test_bnbatch_forward_against_the_float64_train_mode_oracle(which, preset, hm, B):
    """the whole forward: heatmaps within 3 % relative L2 of the float64 oracle run in train mode (about twenty bf16-stored maps deep), per frame;
    every running statistic of the 20 BatchNorm layers against the oracle's; bit-reproducible from the same buffers; parameters untouched"""
    from oracle import hm_ref as H
    net, sd_np = _fresh(which, preset, hm)
    S = 4 * hm
    left, right = _rgb(f"bnbo_L_{which}{hm}", B, S), _rgb(f"bnbo_R_{which}{hm}", B, S)
    sd64 = H.to_torch_sd(sd_np, torch.float64)
    H._TRAIN["on"], H._TRAIN["stats"] = True, {}
    try:
        with torch.no_grad():
            ref = H.hm_forward(left.double(), right.double(), sd64)
        ref_stats = dict(H._TRAIN["stats"])
    finally:
        H._TRAIN["on"] = False
    params0 = {k: v.detach().clone() for k, v in net.named_parameters()}
    bufs0 = {k: v.detach().clone() for k, v in net.named_buffers()}
    out = torch.full((B, 2 * net.num_heatmap + 3, hm, hm), -7.0, device="cuda")      # a wider tensor: the result lands in a channel slice
    net.forward_bnbatch_into(left.cuda(), right.cuda(), out, channel_offset=2, chunk=B)
    torch.cuda.synchronize()
    got = out[:, 2:2 + 2 * net.num_heatmap].double().cpu()
    assert bool((out[:, :2] == -7.0).all()) and bool((out[:, 2 + 2 * net.num_heatmap:] == -7.0).all())
    rels = [float((got[b] - ref[b]).norm() / ref[b].norm()) for b in range(B)]
    print(f"batch-statistics bf16 estimator ({which}, {preset}, {hm}): relative L2 against the float64 train-mode oracle per frame {['%.2e' % r for r in rels]}")
    for b, rel in enumerate(rels):
        assert 1e-5 < rel < 3e-2, (b, rel)
    sd_after = net.state_dict()
    n_checked = 0
    for k, want in ref_stats.items():
        have = sd_after[k].double().cpu()
        rel = float((have - want).norm() / want.norm())
        assert rel < 2e-2, (k, rel)
        n_checked += 1
    assert n_checked == 40                                       # 20 BatchNorm2d layers x (mean, var)
    for k, v in net.named_buffers():
        if k.endswith("num_batches_tracked"):
            assert int(v) == 2, k
    for k, v in net.named_parameters():
        assert torch.equal(v, params0[k]), k
    # the same call from the same buffers: the same bits (outputs and buffers)
    after1 = {k: v.detach().clone() for k, v in net.named_buffers()}
    with torch.no_grad():
        for k, v in net.named_buffers():
            v.copy_(bufs0[k])
    out2 = torch.full_like(out, -7.0)
    net.forward_bnbatch_into(left.cuda(), right.cuda(), out2, channel_offset=2, chunk=B)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    for k, v in net.named_buffers():
        assert torch.equal(v, after1[k]), k


def test_bnbatch_decoder_chunks_and_many_workgroup_batches():
    """B = 40 (640 stem runs on 512 workgroups, the backbone over the whole batch) with the decoder in chunks of 16 (ragged last chunk of 8):
    the BACKBONE maps and statistics do not depend on the chunk size (same bits), the heatmaps only by flipped bf16 roundings (which kernel a
    decoder convolution runs on follows its pixel count); the eval-mode forward from the SAME running statistics differs (another function)."""
    net, _ = _fresh("rot")
    B, hm = 40, 64
    left, right = _rgb("bnbc_L", B, 256).cuda(), _rgb("bnbc_R", B, 256).cuda()
    bufs0 = {k: v.detach().clone() for k, v in net.named_buffers()}
    res = {}
    for chunk in (B, 16):
        with torch.no_grad():
            for k, v in net.named_buffers():
                v.copy_(bufs0[k])
        out = torch.empty((B, 2 * net.num_heatmap, hm, hm), device="cuda")
        net.forward_bnbatch_into(left, right, out, chunk=chunk)
        torch.cuda.synchronize()
        res[chunk] = (out, {n: net.bnbatch_intermediate(n, B, chunk).clone() for n in ("pool0", "layer1", "layer4")},
                      {k: v.detach().clone() for k, v in net.named_buffers()})
    a, b = res[B], res[16]
    for n in a[1]:
        assert torch.equal(a[1][n], b[1][n]), n
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k
    rel = float((a[0].double() - b[0].double()).norm() / a[0].double().norm())
    assert torch.isfinite(a[0]).all() and rel < 1e-2, rel
    with torch.no_grad():
        for k, v in net.named_buffers():
            v.copy_(bufs0[k])
    net.eval()
    ev = net(left, right)
    rel_eval = float((a[0].double() - ev.double()).norm() / ev.double().norm())
    assert rel_eval > 5e-2, rel_eval                             # batch statistics are not the running statistics


def test_bnbatch_refuses_what_it_does_not_cover():
    from egotap_amd import lib
    net, _ = _fresh("pos")
    out = torch.empty((1, 30, 64, 64), device="cuda")
    x = torch.zeros((1, 3, 256, 256), device="cuda")
    with pytest.raises(ValueError, match="two frames"):
        net.forward_bnbatch_into(x, x, out)
    net.set_precision("f32")
    out2, x2 = torch.empty((2, 30, 64, 64), device="cuda"), torch.zeros((2, 3, 256, 256), device="cuda")
    with pytest.raises(lib.EgotapError, match="bf16"):
        net.forward_bnbatch_into(x2, x2, out2)


def test_wrapper_step_from_rgb_under_use_amp_runs_the_batch_statistics_kernels(tmp_path):
    """The reference's stage-2 command line WITHOUT --use_gt_heatmap and WITH --use_amp (scripts/train/PoseEstimator/unrealego.sh passes it):
    create_model(opt) -> model.train() -> optimize_parameters() from RGB.  The frozen estimators take the bf16 channels-last batch-statistics
    forward (models.py forward_heatmap); checked against the REFERENCE WRAPPER'S fp32 fixture of the same step (wrapper_step_rgb_ue_b2.npz) at
    bf16 level: heatmaps 3 % relative L2 on the fixture's strided sample, every running statistic after the step, counters = 2, parameters frozen."""
    from egotap_amd import models, spec
    from test_gpu_wrapper_golden import _opt, _data
    g = np.load(os.path.join(GOLD, "wrapper_step_rgb_ue_b2.npz"))
    lift = {k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(spec.lift_preset("UnrealEgo"))).items()}
    pos = {k: torch.from_numpy(v) for k, v in synth_hm_state_dict(15, "hm_pos.").items()}
    rot = {k: torch.from_numpy(v) for k, v in synth_hm_state_dict(30, "hm_rot.").items()}
    for sub, sd in (("hm_pos", pos), ("hm_sin", rot)):
        os.makedirs(tmp_path / sub)
        torch.save(sd, tmp_path / sub / "best_net_HeatMap.pth")
    opt = _opt(tmp_path, True, False)
    opt.use_amp = True
    m = models.create_model(opt)
    m.net_AutoEncoder.load_state_dict(lift, strict=True)
    m.train()
    frozen0 = {tag: {k: v.clone() for k, v in net.named_parameters()} for tag, net in (("pos", m.net_HeatMap), ("rot", m.net_RotHeatMap))}
    m.set_input(_data(2, "rgbstep"))
    m.optimize_parameters()
    torch.cuda.synchronize()
    assert m.net_HeatMap.precision == "bf16" and m.net_RotHeatMap.precision == "bf16"
    assert getattr(m.net_HeatMap, "_ws_bn", None) is not None                   # the batch-statistics entry point ran (its scratch exists)
    cat = m.pred_heatmap_cat
    have, want = cat.reshape(-1)[::997].double().cpu().numpy(), g["cat_sample_step1"].astype(np.float64)
    rel = float(np.linalg.norm(have - want) / np.linalg.norm(want))
    print(f"--use_amp from RGB, step 1: heatmap sample relative L2 against the reference wrapper's fp32 step {rel:.2e}")
    assert rel < 3e-2, rel
    errs = m.get_current_errors()
    assert list(errs.keys()) == list(g["errors_keys"]) and all(np.isfinite(v) for v in errs.values())
    np.testing.assert_allclose([errs[k] for k in errs], g["errors_step1"], rtol=5e-2, atol=1e-3)
    for tag, net in (("pos", m.net_HeatMap), ("rot", m.net_RotHeatMap)):
        for k, v in net.named_buffers():
            if k.endswith("num_batches_tracked"):
                assert int(v) == 2, (tag, k)
        for k, v in net.named_parameters():
            assert torch.equal(v, frozen0[tag][k]) and not v.requires_grad, k
    # two more steps run and count
    for _ in range(2):
        m.optimize_parameters()
        m.update_learning_rate()
    torch.cuda.synchronize()
    assert int(m.net_RotHeatMap.state_dict()[BB + "bn1.num_batches_tracked"]) == 6
    assert torch.isfinite(m.pred_pose).all()
    # the opt-out keeps the estimators on running statistics: another function of the same input
    opt2 = _opt(tmp_path, True, False)
    opt2.use_amp, opt2.frozen_heatmap_bn_eval = True, True
    m2 = models.create_model(opt2)
    m2.net_AutoEncoder.load_state_dict(lift, strict=True)
    m2.train()
    m2.set_input(_data(2, "rgbstep"))
    m2.optimize_parameters()
    torch.cuda.synchronize()
    assert int(m2.net_HeatMap.state_dict()[BB + "bn1.num_batches_tracked"]) == 0
    assert float((m2.pred_heatmap_cat - cat).abs().max()) > 1e-3
