"""GPU tests of the BASELINE.json configurations that are not the headline:
  config 3  UnrealEgo training step (fwd + bwd + AdamW), bf16, batch 1024, one GPU        -> create_model(opt) under --use_amp
  config 4  the same step data-parallel                                                    -> two ranks sharing this one GPU (gloo)
  config 5  EgoCap preset, 512x512 RGB = 128x128 heatmaps, bf16                            -> forward in plain bf16 + one train step
All of them go through the reference-shaped wrapper (egotap_amd.models.create_model) and the C ABI.  The checker is the float64
oracle (oracle/lift_ref.py, pinned by the reference's golden vectors), never the HIP fp32 path.

bf16 error model used for the gates: an operand rounded to bf16 carries a relative error of at most 2^-9 (round to nearest even,
8 significant bits), products are exact in the fp32 accumulator, so a length-K dot product of rounded operands has a relative error
of about 2^-9 * sqrt(2) on a random-sign sum, independent of K; through the head's ~20 dependent rounded layers the loss and the
large gradient tensors stay within a few 1e-2 (measured: loss 1e-3, gradient cosine >= 0.99).  The gates below are 1e-2 on the
loss, cosine > 0.98 and relative L2 < 0.2 on every large gradient tensor against float64.
"""
import os
import shlex
import socket

import numpy as np
import pytest
import torch

from egotap_amd.synthetic import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# flags of the reference's own stage-2 training script (scripts/train/PoseEstimator/unrealego.sh:4-30), verbatim
UNREALEGO_TRAIN_FLAGS = """
    --project_name UnrealEgoPose --experiment_name egotap_unrealego --model egotap_autoencoder
    --use_amp --init_ImageNet --optimizer_type AdamW --lr_policy cos_anneal_warmup --lr 1e-3
    --gpu_ids 0 --lambda_mpjpe 0.1 --lambda_rot 1.0 --lambda_indep_pos 0.1 --lambda_heatmap 1.0 --lambda_rot_heatmap 1.0
    --lambda_cos_sim -0.01 --lambda_heatmap_rec 0.0 --lambda_rot_heatmap_rec 0.0 --skel_layer PU --ae_hidden_size 128
    --patched_heatmap_ae --niter 1 --niter_decay 15 --batch_size 32 --num_rot_heatmap 15 --num_heatmap 15 --heatmap_type sin
    --path_to_trained_heatmap ./log/unrealego_heatmap_shared/best_net_HeatMap.pth
"""


def _data(B, p, seed="cfg", gt_range=20.0):
    J = p.n_joints_hm
    hm = torch.from_numpy(synth_input(f"hm_{seed}", (min(B, 8), p.in_channels, p.hm_size, p.hm_size)))
    hm = hm.repeat((B + hm.shape[0] - 1) // hm.shape[0], 1, 1, 1)[:B].contiguous()
    gt = torch.from_numpy(synth_input(f"gt_{seed}", (B, p.out_joints, 3), -gt_range, gt_range))
    data = {"input_rgb_left": torch.zeros(1, 3, 4, 4), "input_rgb_right": torch.zeros(1, 3, 4, 4), "gt_heatmap_left": hm[:, :J],
            "gt_heatmap_right": hm[:, J:2 * J], "gt_limb_heatmap_left": hm[:, 2 * J:4 * J], "gt_limb_heatmap_right": hm[:, 4 * J:],
            "gt_local_pose": gt}
    return data, hm, gt


def _model(preset="UnrealEgo", hm=64, **over):
    from egotap_amd import models, spec
    from egotap_amd.options import preset_defaults
    opt = preset_defaults(preset, hm)
    opt.isTrain, opt.use_gt_heatmap, opt.lr, opt.opt_eps, opt.weight_decay = True, True, 1e-3, 1e-4, 0.0
    for k, v in over.items():
        setattr(opt, k, v)
    m = models.create_model(opt)
    p = spec.lift_preset(preset, hm)
    m.net_AutoEncoder.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(p)).items()})
    return m, p


def _oracle_step(hm, gt, p, dtype=torch.float64):
    from egotap_amd import spec
    from oracle import lift_ref as O
    sd = O.to_torch_sd(synth_state_dict(spec.lift_state_spec(p)), dtype)
    return O.train_step(hm.to(dtype), gt.to(dtype), sd, p)


def _grad_gates(net, ref_grads, cos_min, rel_max, min_numel=65536):
    """per large tensor: cosine and relative L2 against the float64 gradient.  Tensors whose gradient is small BY CANCELLATION -- norm
    below 1 % of the largest tensor's: the query / key weights of the deeper layers, whose gradient passes through a near-uniform
    softmax (over 2304 keys at 128 x 128 heatmaps) -- lose relative accuracy in proportion under any bf16 arithmetic (measured: cosine
    0.79-0.96 with fp32 tensors in HBM, 0.82-0.96 with bf16 tensors); they get the loose gate cosine > 0.7."""
    worst = (1.0, 0.0)
    gmax = max(float(g.double().norm()) for g in ref_grads.values() if g is not None)
    for k, v in net.named_parameters():
        g = ref_grads.get(k)
        if g is None:
            assert v.grad is None, k
            continue
        assert v.grad is not None and torch.isfinite(v.grad).all(), k
        if v.numel() < min_numel:
            continue
        a, b = v.grad.double().reshape(-1).cpu(), g.double().reshape(-1)
        if float(b.norm()) < 1e-9:
            continue
        cos = float(a @ b / (a.norm() * b.norm()))
        rel = float((a - b).norm() / b.norm())
        if float(b.norm()) < 1e-2 * gmax:
            assert cos > 0.7, f"{k} (small by cancellation): cos {cos:.5f}"
            continue
        assert cos > cos_min and rel < rel_max, f"{k}: cos {cos:.5f} rel {rel:.3e}"
        worst = (min(worst[0], cos), max(worst[1], rel))
    return worst


# ------------------------------------------------------------------------------------------------------------ --use_amp / flags
def test_create_model_with_the_reference_training_flags(tmp_path):
    """create_model(opt) with the exact flag set of scripts/train/PoseEstimator/unrealego.sh: --use_amp maps to the bf16 mode (no
    GradScaler), --path_to_trained_heatmap loads <dir>_pos/<file> and <dir>_sin/<file> into the two frozen estimators
    (egotap_autoencoder_model.py:113-129), the optimizer is AdamW with cos_anneal_warmup; one optimize_parameters() from RGB runs."""
    from egotap_amd import models, options, spec
    from egotap_amd.synthetic import synth_hm_state_dict
    opt = options.parse_train(shlex.split(UNREALEGO_TRAIN_FLAGS))
    assert opt.use_amp and opt.isTrain and opt.optimizer_type == "AdamW" and opt.lr_policy == "cos_anneal_warmup"
    opt.log_dir = str(tmp_path)
    opt.epoch_iter_cnt = 4
    with pytest.raises(FileNotFoundError):
        models.create_model(opt)                                   # the stage-1 checkpoints are required, as in the reference
    sd_pos = {k: torch.from_numpy(v) for k, v in synth_hm_state_dict(15, "hm_pos.").items()}
    sd_rot = {k: torch.from_numpy(v) for k, v in synth_hm_state_dict(30, "hm_rot.").items()}
    for sub, sd in (("unrealego_heatmap_shared_pos", sd_pos), ("unrealego_heatmap_shared_sin", sd_rot)):
        os.makedirs(tmp_path / sub)
        torch.save(sd, tmp_path / sub / "best_net_HeatMap.pth")
    m = models.create_model(opt)
    for k, v in m.net_HeatMap.state_dict().items():
        assert torch.equal(v.cpu(), sd_pos[k]), k
    for k, v in m.net_RotHeatMap.state_dict().items():
        assert torch.equal(v.cpu(), sd_rot[k]), k
    assert not any(q.requires_grad for q in m.net_HeatMap.parameters()) and not any(q.requires_grad for q in m.net_RotHeatMap.parameters())
    p = spec.lift_preset("UnrealEgo")
    m.net_AutoEncoder.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(p)).items()})
    B = 2
    data = {"input_rgb_left": torch.from_numpy(synth_input("rgb_l", (B, 3, 256, 256), -2.0, 2.0)),
            "input_rgb_right": torch.from_numpy(synth_input("rgb_r", (B, 3, 256, 256), -2.0, 2.0)),
            "gt_local_pose": torch.from_numpy(synth_input("gt_flags", (B, 16, 3), -20.0, 20.0))}
    m.set_input(data)
    m.update_learning_rate()                                       # warm-up starts at lr 0: take one scheduler step first
    before = {k: v.detach().clone() for k, v in m.net_AutoEncoder.named_parameters()}
    m.optimize_parameters()
    assert m.net_AutoEncoder.precision == "bf16"                   # --use_amp -> reduced-precision HIP mode
    assert m.net_HeatMap.precision == "bf16" and m.net_RotHeatMap.precision == "bf16"      # autocast spans the frozen estimators too
    errs = m.get_current_errors()
    assert set(errs) == {"pose", "cos_sim"} and all(np.isfinite(v) for v in errs.values())
    moved = sum(int(not torch.equal(before[k], v.detach())) for k, v in m.net_AutoEncoder.named_parameters())
    assert moved >= 90                                             # every trained tensor moved; cls_token / pooler did not
    # evaluation runs in fp32 whatever --use_amp says (options/test_options.py:15)
    from egotap_amd.training import EgotapAdamW
    assert isinstance(m.optimizers[0], EgotapAdamW)
    class Avg(dict):
        def update(self, d):
            for k, v in d.items():
                self.setdefault(k, []).append(float(v))
    avg = Avg()
    m.eval()                                                       # utils/evaluate.py:150 (every caller of evaluate() in the reference does)
    m.evaluate(avg)
    assert len(avg["mpjpe"]) == B and m.net_AutoEncoder.precision == "bf16"
    # evaluation is fp32 for all three networks: the heatmaps it leaves behind equal an fp32 model's (bit for bit)
    hm_eval = m.pred_heatmap_cat.clone()
    m.set_precision("f32")
    m.forward(evaluate=True)
    assert torch.equal(hm_eval, m.pred_heatmap_cat)
    # the estimators' chunked forward (opt.hm_chunk) equals the unchunked one: frames are independent in eval mode ([r4] to rounding: below 65
    # frames the small-map convolutions split their input channels by the batch they see, conv_f32.h)
    m.opt.hm_chunk = 1
    m.forward(evaluate=True)
    assert float((hm_eval - m.pred_heatmap_cat).abs().max()) < 1e-5 * float(hm_eval.abs().max())

    opt2 = options.parse_train(shlex.split(UNREALEGO_TRAIN_FLAGS.replace("--path_to_trained_heatmap ./log/unrealego_heatmap_shared/best_net_HeatMap.pth", "")))
    with pytest.raises(ValueError):
        models.create_model(opt2)                                  # training from RGB without trained estimators is refused


# ------------------------------------------------------------------------------------------------------------ config 3
def test_config3_bf16_step_b2_against_float64_oracle():
    """one bf16 optimisation step at B = 2 against ONE STEP OF THE FLOAT64 ORACLE (not against the HIP fp32 step)"""
    m, p = _model(use_amp=True)
    data, hm, gt = _data(2, p, "c3", gt_range=1.0)
    m.set_input(data)
    m.optimize_parameters()
    assert m.net_AutoEncoder.precision == "bf16"
    ref = _oracle_step(hm, gt, p)
    errs = m.get_current_errors()
    np.testing.assert_allclose(errs["pose"], float(ref["loss_pose"]), rtol=1e-2)
    np.testing.assert_allclose(errs["cos_sim"], float(ref["loss_cos_sim"]), rtol=5e-2, atol=1e-5)
    pose = m.pred_pose.detach().double().cpu()
    assert float((pose - ref["pose"]).abs().max()) < 3e-2 * float(ref["pose"].abs().max())
    cos, rel = _grad_gates(m.net_AutoEncoder, ref["grads"], 0.98, 0.2)
    print(f"bf16 step vs float64 oracle: worst gradient cosine {cos:.5f}, worst relative L2 {rel:.3e}")


def test_config3_bf16_train_step_b1024_through_the_wrapper():
    """BASELINE config 3 as stated: UnrealEgo, batch 1024, bf16, fwd + bwd + AdamW on one GPU, through create_model(opt).
    Finite losses; a second model stepped on the same batch ends bit-identical (fixed-order reductions); peak memory bounded;
    eval rows of the B = 1024 batch equal the same rows evaluated as a batch of 2 (samples are independent in eval)."""
    B = 1024
    finals, peaks, losses = [], [], []
    for rep in range(2):
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        held = torch.cuda.memory_allocated()        # what earlier tests of this process still hold (cached nets, workspaces)
        m, p = _model(use_amp=True)
        data, hm, gt = _data(B, p, "c3big")
        m.set_input(data)
        m.optimize_parameters()
        torch.cuda.synchronize()
        peaks.append((torch.cuda.max_memory_allocated() - held) / 2 ** 30)
        e = m.get_current_errors()
        losses.append((e["pose"], e["cos_sim"]))
        assert np.isfinite(e["pose"]) and np.isfinite(e["cos_sim"]) and 0.0 < e["pose"] < 10.0
        finals.append({k: v.detach().clone() for k, v in m.net_AutoEncoder.named_parameters()})
        if rep == 1:
            m.net_AutoEncoder.eval()
            with torch.no_grad():
                big = m.net_AutoEncoder.predict_pose(m.pred_heatmap_cat)
                small = m.net_AutoEncoder.predict_pose(m.pred_heatmap_cat[510:512].contiguous())
            assert torch.isfinite(big).all()
            # bf16 rounding of operands is per element, so a row does not depend on its batch; the small batch takes the split-K
            # fp32 path for its few-tile GEMMs, hence a tolerance (bf16-sized) instead of bit equality
            assert float((big[510:512] - small).abs().max()) < 3e-2 * float(small.abs().max())
        del m, data, hm, gt
    assert losses[0] == losses[1]
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k
    print(f"config 3 peak HBM {peaks[0]:.1f} GiB")
    assert peaks[0] < 90.0                  # bf16 activation storage: 79.6 GiB measured (fp32 storage in round 1: 146.7 GiB)


# ------------------------------------------------------------------------------------------------------------ config 5
def test_config5_egocap_hm128_forward_bf16_against_oracle():
    """EgoCap, 128x128 heatmaps (512x512 RGB): 2304 tokens, fc1 K = 65536 / 32768 -- forward in PLAIN bf16 against the float64 oracle"""
    from gpu_util import lift_net
    from oracle import lift_ref as O
    net, sd_np, p = lift_net("EgoCap", 128)
    hm = torch.from_numpy(synth_input("hm_c5", (2, p.in_channels, 128, 128)))
    with torch.no_grad():
        ref = O.lift_forward(hm.double(), O.to_torch_sd(sd_np, torch.float64), p)
    try:
        net.set_precision("bf16")
        got = net.predict_pose(hm.cuda()).double().cpu()
    finally:
        net.set_precision("f32")
    f32 = net.predict_pose(hm.cuda()).double().cpu()
    assert float((f32 - ref).abs().max()) < 1e-4                               # the exact mode: north-star tolerance
    assert float((got - ref).abs().max()) < 3e-2 * float(ref.abs().max())       # bf16 operands: 2^-9 per rounding


def test_config5_egocap_hm128_train_step_against_oracle():
    """one fp32 optimisation step of the EgoCap head at 128x128 heatmaps against oracle.train_step (float64): loss, every gradient
    tensor (5e-3 of its typical magnitude, as the UnrealEgo golden gate), parameters after AdamW"""
    m, p = _model("EgoCap", 128)
    data, hm, gt = _data(2, p, "c5t", gt_range=1.0)
    m.set_input(data)
    m.optimize_parameters()
    ref = _oracle_step(hm, gt, p)
    errs = m.get_current_errors()
    np.testing.assert_allclose(errs["pose"], float(ref["loss_pose"]), rtol=1e-4)
    np.testing.assert_allclose(errs["cos_sim"], float(ref["loss_cos_sim"]), rtol=2e-3, atol=1e-7)
    params = dict(m.net_AutoEncoder.named_parameters())
    for k, g in ref["grads"].items():
        if g is None:
            assert params[k].grad is None, k
            continue
        got = params[k].grad.double().cpu()
        scale = max(float(g.norm()) / np.sqrt(g.numel()), 1e-12)
        err = float((got - g).abs().max())
        # gradients that vanish by symmetry (|g| ~ 1e-12: the final LayerNorm's bias here) are rounding noise of fp32 sums over
        # 2304-token rows: floor 5e-8 (UnrealEgo at 576 tokens: 5e-9)
        assert err <= 5e-3 * scale + 5e-8, f"{k}: err {err:.3e} vs typical magnitude {scale:.3e}"
        np.testing.assert_allclose(params[k].detach().double().cpu().numpy(), ref["new_params"][k].numpy(), atol=3e-5, err_msg=k)
    # and the same step in the bf16 mode tracks it (cosine / relative L2 against float64)
    m2, _ = _model("EgoCap", 128, use_amp=True)
    m2.set_input(data)
    m2.optimize_parameters()
    np.testing.assert_allclose(m2.get_current_errors()["pose"], float(ref["loss_pose"]), rtol=1e-2)
    _grad_gates(m2.net_AutoEncoder, ref["grads"], 0.98, 0.2)


# ------------------------------------------------------------------------------------------------------------ config 4 (2 ranks)
def _ddp_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", EGOTAP_SHARED_DEVICE="1")      # two processes on one GPU: per-step PU kernels
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from egotap_amd import parallel
    torch.cuda.set_device(0)
    parallel.init_from_env("gloo", torch.device("cuda", 0))       # two ranks share the one GPU of the box: gloo, not RCCL
    m, p = _model()
    data, hm, gt = _data(2, p, f"ddp_rank{rank}", gt_range=1.0)
    m.set_input(data)
    m.optimize_parameters()                                        # forward, loss, backward, gradient averaging, AdamW
    torch.cuda.synchronize()
    out = {k: v.grad.detach().cpu() for k, v in m.net_AutoEncoder.named_parameters() if v.grad is not None}
    out.update({"p:" + k: v.detach().cpu() for k, v in m.net_AutoEncoder.named_parameters()})
    torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_config4_two_rank_wrapper_step_on_one_gpu(tmp_path):
    """SURVEY 8(e) parity: two ranks, each the real wrapper on its own shard -> the averaged gradients every rank holds equal the
    gradients of the MEAN of the per-rank losses (BatchNorm statistics stay per rank), i.e. the mean of the two single-process
    gradients; parameters after AdamW are identical on both ranks."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [torch.load(tmp_path / f"rank{r}.pt") for r in range(2)]
    # the same two shards, one process, no collective
    singles = []
    for r in range(2):
        m, p = _model()
        data, hm, gt = _data(2, p, f"ddp_rank{r}", gt_range=1.0)
        m.set_input(data)
        m.net_AutoEncoder.train()
        m.forward()
        m.loss_total = 0.0
        m.backward_AutoEncoder()
        m.loss_total.backward()
        singles.append({k: v.grad.detach().cpu() for k, v in m.net_AutoEncoder.named_parameters() if v.grad is not None})
        del m
    for k in singles[0]:
        mean = (singles[0][k].double() + singles[1][k].double()) / 2
        for r in range(2):
            a = got[r][k].double()
            tol = 1e-6 * float(mean.abs().max()) + 1e-12
            assert float((a - mean).abs().max()) <= tol, (k, r, float((a - mean).abs().max()), tol)
    for k in got[0]:
        assert torch.equal(got[0][k], got[1][k]), k                 # ranks stay in lock step (grads and parameters)


# ------------------------------------------------------------------------------------------------------------ frozen estimators' BatchNorm
def test_frozen_estimators_batchnorm_default_is_the_reference_train_mode_and_the_flag_opts_out(tmp_path):
    """train.py:91 calls model.train() on the wrapper, so the reference's FROZEN estimators normalise with batch statistics and keep
    updating their running statistics while the head trains (egotap_autoencoder_model.py:127-129 freezes parameters only).  That is
    the default here too (pinned against the reference wrapper in test_gpu_wrapper_golden.py); opt.frozen_heatmap_bn_eval opts out:
    eval-mode estimators (what the stage-1 checkpoints were validated with), running statistics and parameters both untouched."""
    from egotap_amd import models, options, spec
    from egotap_amd.hm_training import hm_train_forward_nograd
    from egotap_amd.synthetic import synth_hm_state_dict
    sd_pos = {k: torch.from_numpy(v) for k, v in synth_hm_state_dict(15, "hm_pos.").items()}
    sd_rot = {k: torch.from_numpy(v) for k, v in synth_hm_state_dict(30, "hm_rot.").items()}
    for sub, sd in (("hm_pos", sd_pos), ("hm_sin", sd_rot)):
        os.makedirs(tmp_path / sub)
        torch.save(sd, tmp_path / sub / "best_net_HeatMap.pth")
    B = 2
    data = {"input_rgb_left": torch.from_numpy(synth_input("rgb_l_bn", (B, 3, 256, 256), -2.0, 2.0)),
            "input_rgb_right": torch.from_numpy(synth_input("rgb_r_bn", (B, 3, 256, 256), -2.0, 2.0)),
            "gt_local_pose": torch.from_numpy(synth_input("gt_bn", (B, 16, 3), -20.0, 20.0))}
    key = "backbone.backbone.backbone.bn1.running_mean"
    cats = {}
    for opt_out in (True, False):
        m, p = _model(use_gt_heatmap=False, log_dir=str(tmp_path), path_to_trained_heatmap=str(tmp_path / "hm" / "best_net_HeatMap.pth"),
                      frozen_heatmap_bn_eval=opt_out)
        m.train()                                                             # train.py:91
        assert m.net_HeatMap.training and m.net_RotHeatMap.training
        before = {k: v.clone() for k, v in m.net_HeatMap.state_dict().items()}
        m.set_input(data)
        m.optimize_parameters()
        after = m.net_HeatMap.state_dict()
        moved = not torch.equal(before[key], after[key])
        assert moved == (not opt_out)
        assert m.net_HeatMap.training and m.net_RotHeatMap.training           # the forward leaves the modes alone
        for k, v in m.net_HeatMap.named_parameters():
            assert torch.equal(before[k], after[k]) and not v.requires_grad      # frozen either way
        cats[opt_out] = m.pred_heatmap_cat.clone()
        if not opt_out:
            # the heatmaps the head trained on are the train-mode forward (batch statistics per eye) of the estimators as they were
            ref = models.create_model(m.opt)
            ref.net_HeatMap.load_state_dict(before)
            ref.net_HeatMap.train()
            J = p.n_joints_hm
            want = hm_train_forward_nograd(ref.net_HeatMap, m.input_rgb_left.float().contiguous(), m.input_rgb_right.float().contiguous())
            assert torch.equal(cats[False][:, :2 * J], want)
        else:
            m.eval()
            m.forward()
            assert torch.equal(m.pred_heatmap_cat, cats[True])                # opted out: model.train() / model.eval() give the same heatmaps
    assert not torch.equal(cats[False], cats[True])


def test_bf16_inference_split_k_fc1_equals_the_unsplit_rows():
    """bf16-storage inference with few encoder rows: fc1 of both encoders (K = 16384 / 8192, N = 2048) splits K over the chip when its
    row tiles would leave most CUs idle (B = 8: 240 rows = 8 tiles -> 32 splits) and sums the fp32 partial slabs in a fixed order.  Rows
    are batch independent in eval, so the same frames inside a batch of 256 (240 tiles: no split) must give the same poses up to the
    fp32 summation order of that one product; run to run bit-identical."""
    from gpu_util import lift_net
    net, _, p = lift_net("UnrealEgo")
    hm = torch.from_numpy(synth_input("hm_splitk", (8, p.in_channels, 64, 64))).cuda()
    big = hm.repeat(32, 1, 1, 1).contiguous()
    try:
        net.set_precision("bf16")
        small = net.predict_pose(hm).clone()
        again = net.predict_pose(hm).clone()
        full = net.predict_pose(big)[:8].clone()
    finally:
        net.set_precision("f32")
    assert torch.equal(small, again)
    scale = float(full.abs().max())
    err = float((small - full).abs().max())
    print(f"split-K fc1 (B = 8) vs unsplit rows (B = 256): max |diff| {err:.3e} at pose scale {scale:.3f}")
    assert err < 2e-4 * scale


def test_bf16_inference_rows_do_not_depend_on_the_batch_size():
    """[r4] bf16 inference takes the bf16-storage kernels from 512 token rows on (B = 1 at 576 tokens per frame): every ViT product of a
    B = 1 / 2 / 4 forward is then the SAME kernel in the same k order as in a batch of 256 (no split-K inside the ViT: a split there would
    re-order fp32 sums in front of a LayerNorm -> bf16 rounding, and poses would move by 3e-3 with the batch size -- measured, not adopted);
    only fc1's few-row split-K re-orders one fp32 product.  A frame's pose in a batch of 1, 2 or 4 = its pose in a batch of 256 to 2e-4."""
    from gpu_util import lift_net
    net, _, p = lift_net("UnrealEgo")
    hm = torch.from_numpy(synth_input("hm_splitk", (8, p.in_channels, 64, 64))).cuda()
    big = hm.repeat(32, 1, 1, 1).contiguous()
    try:
        net.set_precision("bf16")
        full = net.predict_pose(big)[:8].clone()
        parts = {n: net.predict_pose(hm[:n].contiguous()).clone() for n in (1, 2, 4)}
    finally:
        net.set_precision("f32")
    scale = float(full.abs().max())
    for n, out in parts.items():
        err = float((out - full[:n]).abs().max())
        print(f"bf16 inference, batch of {n} vs the same frames in a batch of 256: max |diff| {err:.3e} at pose scale {scale:.3f}")
        assert err < 2e-4 * scale, (n, err)
