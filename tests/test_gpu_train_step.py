"""GPU parity of one whole training step of the lifting head (forward in train mode, loss, backward, AdamW -- all
HIP, PyTorch only as autograd glue) against golden vectors captured from ONE optimisation step of the reference's own
modules (tools/make_golden.py gen_train) and against the reference wrapper's optimize_parameters() semantics."""
import os

import numpy as np
import pytest
import torch

from egotap_amd.synthetic import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _fresh_net():
    from egotap_amd import networks, spec
    from egotap_amd.options import preset_defaults
    p = spec.lift_preset("UnrealEgo")
    net = networks.EgoTAPAutoEncoder(preset_defaults("UnrealEgo"), input_channel_scale=2)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(p)).items()})
    return net.cuda(), p


def _check_against_golden(net, g, losses, abs_floor=1e-8):
    # abs_floor: gradients that vanish by symmetry in the reference (|g| ~ 5e-10: the final LayerNorm's bias) are rounding noise of
    # fp32 sums; its size depends on the order of the additions (the two loss terms' gradients are added outside the kernel now)
    norms = dict(zip(g["grad_keys"], g["grad_norms"]))
    params = dict(net.named_parameters())
    assert sorted(k for k, v in params.items() if v.grad is not None) == sorted(g["grad_keys"])
    assert sorted(k for k, v in params.items() if v.grad is None) == sorted(g["no_grad_keys"])
    np.testing.assert_allclose(losses[0], g["loss_pose"], rtol=1e-4)
    np.testing.assert_allclose(losses[1], g["loss_cos_sim"], rtol=2e-3, atol=1e-7)
    worst = 0.0
    for k in g["grad_keys"]:
        gr = params[k].grad
        got = gr.reshape(-1)[:: max(1, gr.numel() // 257)].cpu().numpy()
        scale = max(norms[k] / np.sqrt(gr.numel()), 1e-12)
        err = np.abs(got - g["g:" + k]).max()
        assert err <= 5e-3 * scale + abs_floor, f"{k}: sample err {err:.3e} vs typical magnitude {scale:.3e}"
        np.testing.assert_allclose(float(gr.double().norm()), norms[k], rtol=1e-3, atol=5e-9 * np.sqrt(gr.numel()) + 1e-8, err_msg=k)  # zero-by-symmetry gradients are noise
        worst = max(worst, err / scale)
    return worst


def test_train_forward_backward_adamw_match_reference():
    from egotap_amd.training import EgotapAdamW, PoseLossFn
    g = np.load(os.path.join(GOLD, "train_step_ue_b2.npz"))
    net, p = _fresh_net()
    net.train()
    hm = torch.from_numpy(synth_input("hm_train", (2, 90, 64, 64))).cuda()
    gt = torch.from_numpy(synth_input("gt_train", (2, 16, 3), -1.0, 1.0)).cuda()
    opt = EgotapAdamW(net.parameters(), lr=1e-3, eps=1e-4, weight_decay=0.0)
    opt.zero_grad()
    pose = net(hm)[0]
    assert pose.requires_grad
    np.testing.assert_allclose(pose.detach().cpu().numpy(), g["pose"], atol=1e-4)
    both = PoseLossFn.apply(net, pose, gt, 0.1, -0.01)
    both.sum().backward()
    torch.cuda.synchronize()
    _check_against_golden(net, g, both.detach().cpu().numpy())
    sd = net.state_dict()
    for k in g.files:
        if k.startswith("buf:"):
            np.testing.assert_allclose(sd[k[4:]].cpu().numpy(), g[k], atol=2e-6, err_msg=k)
    opt.step()
    torch.cuda.synchronize()
    params = dict(net.named_parameters())
    for k in g["grad_keys"]:
        got = params[k].detach().reshape(-1)[:: max(1, params[k].numel() // 257)].cpu().numpy()
        np.testing.assert_allclose(got, g["p:" + k], atol=3e-5, err_msg=k)
    for k in g["no_grad_keys"]:        # cls_token / pooler never move
        assert params[k].grad is None


def test_training_step_is_deterministic_and_eval_unchanged():
    net, p = _fresh_net()
    hm = torch.from_numpy(synth_input("hm_train", (3, 90, 64, 64))).cuda()
    gt = torch.from_numpy(synth_input("gt_train3", (3, 16, 3), -1.0, 1.0)).cuda()
    from egotap_amd.training import PoseLossFn
    grads = []
    for _ in range(2):
        net.train()
        net.zero_grad()
        PoseLossFn.apply(net, net(hm)[0], gt, 0.1, -0.01).sum().backward()
        grads.append({k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None})
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k        # fixed-order reductions: bitwise reproducible
    net.eval()
    with torch.no_grad():
        assert net(hm)[0].requires_grad is False


def test_wrapper_optimize_parameters():
    """create_model(opt) with isTrain: set_input -> optimize_parameters -> get_current_errors, as train.py drives it"""
    from egotap_amd import models, spec
    from egotap_amd.options import preset_defaults
    g = np.load(os.path.join(GOLD, "train_step_ue_b2.npz"))
    opt = preset_defaults("UnrealEgo")
    opt.isTrain, opt.use_gt_heatmap, opt.lr, opt.opt_eps, opt.weight_decay = True, True, 1e-3, 1e-4, 0.0
    m = models.create_model(opt)
    p = spec.lift_preset("UnrealEgo")
    m.net_AutoEncoder.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(p)).items()})
    hm = torch.from_numpy(synth_input("hm_train", (2, 90, 64, 64)))
    data = {"input_rgb_left": torch.zeros(2, 3, 256, 256), "input_rgb_right": torch.zeros(2, 3, 256, 256),
            "gt_heatmap_left": hm[:, :15], "gt_heatmap_right": hm[:, 15:30], "gt_limb_heatmap_left": hm[:, 30:60],
            "gt_limb_heatmap_right": hm[:, 60:], "gt_local_pose": torch.from_numpy(synth_input("gt_train", (2, 16, 3), -1.0, 1.0))}
    m.set_input(data)
    m.optimize_parameters()
    errs = m.get_current_errors()
    np.testing.assert_allclose(errs["pose"], g["loss_pose"], rtol=1e-4)
    np.testing.assert_allclose(errs["cos_sim"], g["loss_cos_sim"], rtol=2e-3, atol=1e-7)
    params = dict(m.net_AutoEncoder.named_parameters())
    for k in list(g["grad_keys"])[::7]:
        got = params[k].detach().reshape(-1)[:: max(1, params[k].numel() // 257)].cpu().numpy()
        np.testing.assert_allclose(got, g["p:" + k], atol=3e-5, err_msg=k)


def _one_step(mode, B=2):
    from egotap_amd.training import PoseLossFn
    net, p = _fresh_net()
    net.train()
    net.set_precision(mode)
    hm = torch.from_numpy(synth_input("hm_train", (B, 90, 64, 64))).cuda()
    gt = torch.from_numpy(synth_input("gt_train", (B, 16, 3), -1.0, 1.0)).cuda()
    net.zero_grad()
    pose = net(hm)[0]
    both = PoseLossFn.apply(net, pose, gt, 0.1, -0.01)
    both.sum().backward()
    torch.cuda.synchronize()
    return net, pose.detach(), both.detach().cpu().numpy()


def test_train_step_bf16x3_mode_matches_reference_golden():
    """the split-bf16 kernels (forward, input- and weight-gradient GEMMs) keep fp32-grade gradients: same golden, same gates"""
    g = np.load(os.path.join(GOLD, "train_step_ue_b2.npz"))
    net, pose, losses = _one_step("bf16x3")
    np.testing.assert_allclose(pose.cpu().numpy(), g["pose"], atol=1e-4)
    # per-tensor gates inside (5e-3 of the tensor's typical magnitude); gradients that vanish by symmetry (|g| ~ 5e-10) are
    # rounding noise, whose floor is 2^-16-grade here
    _check_against_golden(net, g, losses, abs_floor=3e-8)


def test_train_step_bf16_mode_against_float64_oracle():
    """BASELINE configs 3-4 arithmetic (bf16 MFMA, fp32 accumulate, fp32 master weights) against ONE STEP OF THE FLOAT64 ORACLE on
    the golden's inputs: loss within 1 %, every large gradient tensor within 20 % relative L2 and cosine > 0.98 (bf16 keeps 8
    significant bits per operand; the deepest gradient, the position embeddings behind three transformer layers, is the worst).
    The wrapper-level version of this test (create_model under --use_amp) is tests/test_gpu_configs.py."""
    from egotap_amd import spec
    from oracle import lift_ref as O
    net, pose16, l16 = _one_step("bf16")
    p = spec.lift_preset("UnrealEgo")
    hm = torch.from_numpy(synth_input("hm_train", (2, 90, 64, 64))).double()
    gt = torch.from_numpy(synth_input("gt_train", (2, 16, 3), -1.0, 1.0)).double()
    ref = O.train_step(hm, gt, O.to_torch_sd(synth_state_dict(spec.lift_state_spec(p)), torch.float64), p)
    np.testing.assert_allclose(l16[0], float(ref["loss_pose"]), rtol=1e-2)
    assert float((pose16.double().cpu() - ref["pose"]).abs().max()) < 5e-2 * float(ref["pose"].abs().max())
    from test_gpu_configs import _grad_gates          # one definition of the bf16 gradient gates (incl. the small-by-cancellation rule)
    _grad_gates(net, ref["grads"], 0.98, 0.2)


def test_wrapper_checkpoint_roundtrip_and_scheduler(tmp_path):
    """save_networks / load_networks / load_optimizers with the reference's file naming (base_model.py:64-148) and the
    cos_anneal_warmup schedule driven by update_learning_rate (train.py:130): a resumed model continues bit for bit."""
    from egotap_amd import models, spec
    from egotap_amd.options import preset_defaults

    def make():
        opt = preset_defaults("UnrealEgo")
        opt.isTrain, opt.use_gt_heatmap, opt.lr, opt.opt_eps, opt.weight_decay = True, True, 1e-3, 1e-4, 0.0
        opt.lr_policy, opt.niter, opt.niter_decay, opt.epoch_iter_cnt, opt.epoch_count = "cos_anneal_warmup", 1, 3, 2, 1
        opt.log_dir, opt.experiment_name = str(tmp_path), "exp"
        return models.create_model(opt)
    p = spec.lift_preset("UnrealEgo")
    hm = torch.from_numpy(synth_input("hm_train", (2, 90, 64, 64)))
    data = {"input_rgb_left": torch.zeros(2, 3, 256, 256), "input_rgb_right": torch.zeros(2, 3, 256, 256),
            "gt_heatmap_left": hm[:, :15], "gt_heatmap_right": hm[:, 15:30], "gt_limb_heatmap_left": hm[:, 30:60],
            "gt_limb_heatmap_right": hm[:, 60:], "gt_local_pose": torch.from_numpy(synth_input("gt_train", (2, 16, 3), -1.0, 1.0))}
    a = make()
    a.net_AutoEncoder.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(p)).items()})
    a.set_input(data)
    lrs = []
    for _ in range(3):
        lrs.append(a.optimizers[0].param_groups[0]["lr"])
        a.optimize_parameters()
        a.update_learning_rate()
    np.testing.assert_allclose(lrs, [0.0, 5e-4, 1e-3], rtol=1e-12)            # linear warm-up over niter * epoch_iter_cnt = 2 steps
    a.save_networks(which_epoch=1)
    a.save_networks(which_epoch=2)                                             # numbered epochs: the previous one is removed
    names = sorted(os.listdir(a.save_dir))
    assert names == sorted(["2_net_HeatMap.pth", "2_net_RotHeatMap.pth", "2_net_AutoEncoder.pth", "2_optim_0.pth", "2_scheduler_0.pth"]), names
    sd = torch.load(os.path.join(a.save_dir, "2_net_AutoEncoder.pth"))
    assert list(sd.keys()) == [k for k, _ in spec.lift_state_spec(p)]          # the reference's keys, in its order
    b = make()
    b.load_networks(which_epoch=2)
    b.load_optimizers(which_epoch=2)
    b.set_input(data)
    for m in (a, b):
        m.optimize_parameters()
        m.update_learning_rate()
    assert a.optimizers[0].param_groups[0]["lr"] == b.optimizers[0].param_groups[0]["lr"]
    for (k, x), (_, y) in zip(a.net_AutoEncoder.state_dict().items(), b.net_AutoEncoder.state_dict().items()):
        assert torch.equal(x, y), k
    # a torch.optim.AdamW state file (step kept as a tensor) resumes too
    ref_opt = torch.optim.AdamW([torch.nn.Parameter(v.detach().clone()) for v in a.net_AutoEncoder.parameters()], lr=1e-3, eps=1e-4, weight_decay=0.0)
    for prm in ref_opt.param_groups[0]["params"]:
        prm.grad = torch.zeros_like(prm)
    ref_opt.step()
    b.optimizers[0].load_state_dict(ref_opt.state_dict())
    b.optimize_parameters()


@pytest.mark.parametrize("preset,B,mode", [("UnrealEgo", 3, "f32"), ("EgoCap", 2, "f32"), ("UnrealEgo", 3, "bf16"), ("EgoCap", 2, "bf16x3")])
def test_one_call_training_abi_equals_the_operator_composition(preset, B, mode):
    """egotap_lift_forward_train + egotap_lift_backward (the one-call ABI the wrapper trains through) against the same step composed
    operator by operator from Python (net.one_call_training = False): pose, every gradient and the BatchNorm running statistics are
    bit-identical -- the library composes the same launches in the same order -- and num_batches_tracked advances"""
    from egotap_amd import networks, spec
    from egotap_amd.options import preset_defaults
    from egotap_amd.training import PoseLossFn
    p = spec.lift_preset(preset)
    hm = torch.from_numpy(synth_input(f"hm_onecall_{preset}", (B, p.in_channels, 64, 64))).cuda()
    gt = torch.from_numpy(synth_input(f"gt_onecall_{preset}", (B, p.out_joints, 3), -1.0, 1.0)).cuda()
    out = []
    for one_call in (True, False):
        net = networks.EgoTAPAutoEncoder(preset_defaults(preset), input_channel_scale=2)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(p)).items()})
        net = net.cuda().train()
        net.set_precision(mode)
        net.one_call_training = one_call
        pose = net(hm)[0]
        PoseLossFn.apply(net, pose, gt, 0.1, -0.01).sum().backward()
        torch.cuda.synchronize()
        grads = {k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None}
        bufs = {k: v.clone() for k, v in net.named_buffers()}
        out.append((pose.detach().clone(), grads, bufs))
    (pose_a, g_a, b_a), (pose_b, g_b, b_b) = out
    assert torch.equal(pose_a, pose_b)
    assert sorted(g_a) == sorted(g_b) and len(g_a) > 50
    for k in g_a:
        assert torch.equal(g_a[k], g_b[k]), k
    for k in b_a:
        assert torch.equal(b_a[k], b_b[k]), k
    assert int(b_a["pos_heatmap_encoder.fc1.bn.num_batches_tracked"]) == 1


@pytest.mark.parametrize("one_call", [True, False])
def test_second_backward_before_zero_grad_accumulates(one_call):
    """.grad is a view of the flat gradient arena, which every backward overwrites: a second backward before zero_grad() must ADD to
    the gradient the parameters hold (gradient accumulation, several losses), not replace it"""
    from egotap_amd.training import PoseLossFn
    net, p = _fresh_net()
    net.train()
    net.one_call_training = one_call
    hm = torch.from_numpy(synth_input("hm_train", (2, 90, 64, 64))).cuda()
    gts = [torch.from_numpy(synth_input(f"gt_acc{i}", (2, 16, 3), -1.0, 1.0)).cuda() for i in range(2)]
    bufs = {k: v.clone() for k, v in net.named_buffers()}
    singles = []
    for gt in gts:
        for k, v in net.named_buffers():
            v.copy_(bufs[k])
        net.zero_grad()
        PoseLossFn.apply(net, net(hm)[0], gt, 0.1, -0.01).sum().backward()
        singles.append({k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None})
    net.zero_grad()
    for gt in gts:                                               # two backwards, no zero_grad in between
        for k, v in net.named_buffers():
            v.copy_(bufs[k])
        PoseLossFn.apply(net, net(hm)[0], gt, 0.1, -0.01).sum().backward()
    torch.cuda.synchronize()
    for k, v in net.named_parameters():
        if v.grad is not None:
            assert torch.equal(v.grad, singles[0][k] + singles[1][k]), k


def test_no_grad_train_forward_does_not_keep_a_second_activation_buffer():
    """a train-mode forward whose backward never runs (no_grad, an exception) leaves the activations' buffer marked busy; the next
    forward must release it before allocating its replacement (30 GB each at B = 1024)"""
    net, p = _fresh_net()
    net.train()
    hm = torch.from_numpy(synth_input("hm_train", (2, 90, 64, 64))).cuda()
    with torch.no_grad():
        net(hm)
    first = net._saved_pool["buf"]
    ptr, size = first.data_ptr(), first.numel()
    del first
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        net(hm)
    torch.cuda.synchronize()
    assert net._saved_pool["buf"].numel() == size
    assert torch.cuda.max_memory_allocated() - base < size // 2      # never two activation buffers alive at once


def test_one_call_backward_on_a_one_rank_rccl_group():
    """RCCL on the box's one GPU: a world-size-1 "nccl" process group, the reducer forced on -- the one-call backward records its
    bucket events, every bucket is all-reduced (ReduceOp.AVG) on the side stream behind its event, finish() orders the compute
    stream behind them.  Averaging over one rank is the identity: gradients equal the no-collective step bit for bit."""
    import socket
    import torch.distributed as dist
    from egotap_amd.training import PoseLossFn
    hm = torch.from_numpy(synth_input("hm_train", (2, 90, 64, 64))).cuda()
    gt = torch.from_numpy(synth_input("gt_train", (2, 16, 3), -1.0, 1.0)).cuda()
    net, p = _fresh_net()
    net.train()
    bufs = {k: v.clone() for k, v in net.named_buffers()}
    PoseLossFn.apply(net, net(hm)[0], gt, 0.1, -0.01).sum().backward()
    want = {k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None}
    assert not dist.is_initialized()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        red = net._reducer()
        red.force = True
        for k, v in net.named_buffers():
            v.copy_(bufs[k])
        net.zero_grad()
        PoseLossFn.apply(net, net(hm)[0], gt, 0.1, -0.01).sum().backward()
        torch.cuda.synchronize()
        assert red.last_buckets == p.vit_layers + 2 and red.steps == 1
        n_params = sum(v.numel() for v in net.parameters() if v.grad is not None)
        assert 4 * n_params <= red.last_bytes <= 4 * n_params + 256 * 200      # whole arena (256-byte aligned slices), once
        assert red.read_exposed_ms() >= 0.0
        for k, v in net.named_parameters():
            if v.grad is not None:
                assert torch.equal(v.grad, want[k]), k
    finally:
        net._reducer().force = False
        dist.destroy_process_group()
