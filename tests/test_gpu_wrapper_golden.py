"""egotap_amd.models.EgoTAPAutoEncoderModel against fixtures produced by the REFERENCE'S OWN WRAPPER
(model/egotap_autoencoder_model.py:155-237, 284-350; tools/make_golden.py gen_wrapper): three optimize_parameters() with the
shipped training flags (frozen estimators from <dir>_pos / <dir>_sin checkpoints, --use_gt_heatmap, AdamW, cos_anneal_warmup,
update_learning_rate between steps) and evaluate() of the test-mode wrapper after load_networks('best')."""
import os

import numpy as np
import pytest
import torch

from egotap_amd.synthetic import synth_hm_state_dict, synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _opt(tmp, is_train, use_gt_heatmap):
    """the same flag values tools/make_golden.py::_wrapper_opt gave the reference"""
    from egotap_amd.options import preset_defaults
    opt = preset_defaults("UnrealEgo")
    opt.model, opt.isTrain, opt.use_amp, opt.gpu_ids = "egotap_autoencoder", is_train, False, [0]
    opt.log_dir, opt.experiment_name = str(tmp), "gold_wrapper"
    opt.use_gt_heatmap = use_gt_heatmap
    opt.path_to_trained_heatmap = "hm/best_net_HeatMap.pth" if is_train else None
    opt.optimizer_type, opt.lr, opt.opt_eps, opt.weight_decay, opt.lr_policy = "AdamW", 1e-3, 1e-4, 0.0, "cos_anneal_warmup"
    opt.niter, opt.niter_decay, opt.epoch_iter_cnt, opt.epoch_count = 1, 15, 4, 1
    opt.lambda_mpjpe, opt.lambda_cos_sim = 0.1, -0.01
    return opt


def _data(B, tag):
    hm = torch.from_numpy(synth_input(f"wrap_hm_{tag}", (B, 90, 64, 64)))
    return {"input_rgb_left": torch.from_numpy(synth_input(f"wrap_rgbL_{tag}", (B, 3, 256, 256), -2.0, 2.0)),
            "input_rgb_right": torch.from_numpy(synth_input(f"wrap_rgbR_{tag}", (B, 3, 256, 256), -2.0, 2.0)),
            "gt_heatmap_left": hm[:, :15], "gt_heatmap_right": hm[:, 15:30],
            "gt_limb_heatmap_left": hm[:, 30:60], "gt_limb_heatmap_right": hm[:, 60:],
            "gt_local_pose": torch.from_numpy(synth_input(f"wrap_gt_{tag}", (B, 16, 3), -20.0, 20.0)),
            "gt_local_rot": torch.zeros(B, 16, 3), "gt_limb_theta": torch.zeros(B, 15),
            "gt_pelvis_left": torch.zeros(B, 3), "gt_pelvis_right": torch.zeros(B, 3),
            "gt_plength_left": torch.ones(B, 30), "gt_plength_right": torch.ones(B, 30)}


def _state_dicts():
    from egotap_amd import spec
    lift = {k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(spec.lift_preset("UnrealEgo"))).items()}
    pos = {k: torch.from_numpy(v) for k, v in synth_hm_state_dict(15, "hm_pos.").items()}
    rot = {k: torch.from_numpy(v) for k, v in synth_hm_state_dict(30, "hm_rot.").items()}
    return lift, pos, rot


def _strided(t):
    return t.detach().reshape(-1)[:: max(1, t.numel() // 257)].cpu().numpy()


SPREAD = np.load(os.path.join(GOLD, "wrapper_step_spread.npz"))


def _spread(key):
    """how far the REFERENCE moves from itself on this quantity when only its CPU thread count (summation order) changes: max over 1 and 3
    threads against the fixtures' 8 (tools/make_golden.py gen_wrapper_spread)"""
    return max(float(SPREAD[f"{key[0]}_t{t}_{key[1]}"].max()) for t in (1, 3))


def _close(label, got, want, atol, spread_key=None):
    """assert_allclose(atol) that also PRINTS the observed distance next to the gate and, where recorded, the reference's own spread"""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err = float(np.abs(got - want).max())
    ref = f", reference's own thread-count spread {_spread(spread_key):.1e}" if spread_key else ""
    print(f"{label}: max |gpu - reference| = {err:.2e} (gate {atol:.0e}{ref})")
    assert err <= atol, (label, err, atol)


def test_three_wrapper_steps_match_the_reference_wrapper(tmp_path):
    from egotap_amd import models
    g = np.load(os.path.join(GOLD, "wrapper_step_ue_b2.npz"))
    lift, pos, rot = _state_dicts()
    for sub, sd in (("hm_pos", pos), ("hm_sin", rot)):
        os.makedirs(tmp_path / sub)
        torch.save(sd, tmp_path / sub / "best_net_HeatMap.pth")
    m = models.create_model(_opt(tmp_path, True, True))
    m.net_AutoEncoder.load_state_dict(lift, strict=True)
    assert list(m.loss_names) == list(g["loss_names"])
    # frozen estimators: loaded from the two checkpoints, no parameter trains (egotap_autoencoder_model.py:113-129)
    assert [int(any(q.requires_grad for q in n.parameters())) for n in (m.net_HeatMap, m.net_RotHeatMap)] == list(g["frozen_requires_grad"])
    assert torch.equal(m.net_HeatMap.state_dict()["after_backbone.conv_heatmap.weight"].cpu(), pos["after_backbone.conv_heatmap.weight"])
    assert m.optimizers[0].param_groups[0]["lr"] == float(g["lr_before"][0])            # warm-up starts at zero
    m.set_input(_data(2, "step"))
    params = dict(m.net_AutoEncoder.named_parameters())
    norms = dict(zip(g["grad_keys"], g["grad_norms"]))

    # ---- step 1 (lr = 0: gradients and BatchNorm statistics move, parameters do not).  Run as the operator-by-operator composition
    # (bit-identical to the one-call ABI) with the LeakyReLU branch of every encoder row recorded: see the EgoCap test below
    from egotap_amd import training
    from egotap_amd import train_ops as T
    taken, real = [], T.bn_lrelu_bwd

    def spy(z, y, *a, **k):
        taken.append((y > 0).cpu())
        return real(z, y, *a, **k)
    m.net_AutoEncoder.one_call_training = False
    training.T.bn_lrelu_bwd = spy
    try:
        m.optimize_parameters()
        torch.cuda.synchronize()
    finally:
        training.T.bn_lrelu_bwd = real
        m.net_AutoEncoder.one_call_training = True
    errs = m.get_current_errors()
    assert list(errs.keys()) == list(g["errors_keys"])
    np.testing.assert_allclose([errs[k] for k in errs], g["errors_step1"], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(m.pred_pose.detach().cpu().numpy(), g["pred_pose_step1"], atol=1e-4)
    cat = m.pred_heatmap_cat
    np.testing.assert_allclose(cat.reshape(-1)[::997].cpu().numpy(), g["pred_heatmap_cat_sample"], atol=0, rtol=0)      # the channel order of the concat
    np.testing.assert_allclose([float(cat.double().sum()), float(cat.double().abs().sum())], g["pred_heatmap_cat_stats"], rtol=1e-9)
    shapes = [list(getattr(m, n).shape) for n in ("pred_heatmap_left_rec", "pred_heatmap_right_rec", "pred_limb_heatmap_left_rec",
                                                  "pred_limb_heatmap_right_rec")]
    assert shapes == g["rec_shapes"].tolist() and float(m.pred_heatmap_rec_cat.abs().sum()) == float(g["rec_abs_sum"][0])
    assert list(m.pred_rot.shape) == list(g["pred_rot_shape"]) and list(m.pred_indep_pos.shape) == list(g["pred_indep_pos_shape"])
    assert sorted(k for k, v in params.items() if v.grad is None) == sorted(g["no_grad_keys"])
    assert sorted(k for k, v in params.items() if v.grad is not None) == sorted(g["grad_keys"])
    # Gradients: against the reference's own numbers where every LeakyReLU input of the step took the branch the float64 oracle takes;
    # if a later kernel change flips one (an input within ~1e-5 of zero: DESIGN section 5), against the oracle on the branches taken
    from oracle import lift_ref as O
    from egotap_amd import spec
    p_ = spec.lift_preset("UnrealEgo")
    order = [f"{e}_heatmap_encoder.fc{j}" for e in ("rot", "pos") for j in (3, 2, 1)]
    assert len(taken) == 6
    hook = {"masks": dict(zip(order, taken)), "pre": {}}
    d = _data(2, "step")
    hm_cat = torch.cat([d["gt_heatmap_left"], d["gt_heatmap_right"], d["gt_limb_heatmap_left"], d["gt_limb_heatmap_right"]], 1)
    sd64 = O.to_torch_sd({k: v.numpy() for k, v in lift.items()}, torch.float64)
    want = O.train_step(hm_cat.double(), d["gt_local_pose"].double(), sd64, p_, lrelu=hook)["grads"]
    flips = sum(int((hook["masks"][k] != (hook["pre"][k] > 0)).sum()) for k in order)
    for k in order:
        differ = hook["masks"][k] != (hook["pre"][k] > 0)
        assert not differ.any() or float(hook["pre"][k][differ].abs().max()) < 1e-4, k
    print(f"LeakyReLU branches that differ from the float64 oracle's: {flips}")
    for k in g["grad_keys"]:
        gr = params[k].grad
        scale = max(norms[k] / np.sqrt(gr.numel()), 1e-12)
        if flips == 0:
            err = np.abs(_strided(gr) - g["g:" + k]).max()
            assert err <= 5e-3 * scale + 1e-8, f"{k}: sample err {err:.3e} vs typical magnitude {scale:.3e}"
            np.testing.assert_allclose(float(gr.double().norm()), norms[k], rtol=1e-3, atol=5e-9 * np.sqrt(gr.numel()) + 1e-8, err_msg=k)
        else:
            err = float((gr.double().cpu() - want[k]).abs().max())
            assert err <= 5e-3 * scale + 1e-8, f"{k}: err {err:.3e} vs typical magnitude {scale:.3e} (oracle on the HIP branches)"
        np.testing.assert_allclose(_strided(params[k]), g["p1:" + k], atol=0, rtol=0, err_msg=k)        # lr 0: bit-unchanged
    m.update_learning_rate()
    np.testing.assert_allclose(m.optimizers[0].param_groups[0]["lr"], g["lr_after_step1"][0], rtol=1e-12)

    # ---- step 2 (same parameters -> same losses; first real update)
    m.optimize_parameters()
    torch.cuda.synchronize()
    errs2 = m.get_current_errors()
    np.testing.assert_allclose([errs2[k] for k in errs2], g["errors_step2"], rtol=2e-4, atol=1e-7)
    _close("gt step 2 pose", m.pred_pose.detach().cpu().numpy(), g["pred_pose_step2"], 1e-4, ("gt", "pose_step2"))
    for k, want in zip(g["grad_keys"], g["param_sums_step2"]):
        np.testing.assert_allclose(_strided(params[k]), g["p2:" + k], atol=3e-5, err_msg=k)
        np.testing.assert_allclose(float(params[k].detach().double().sum()), want, rtol=1e-4, atol=3e-5 * np.sqrt(params[k].numel()) + 1e-4, err_msg=k)
    m.update_learning_rate()
    np.testing.assert_allclose(m.optimizers[0].param_groups[0]["lr"], g["lr_after_step2"][0], rtol=1e-12)

    # ---- step 3: the forward now runs on updated parameters (AdamW's first step moves every weight by ~lr)
    m.optimize_parameters()
    torch.cuda.synchronize()
    errs3 = m.get_current_errors()
    np.testing.assert_allclose([errs3[k] for k in errs3], g["errors_step3"], rtol=2e-3, atol=2e-5)
    # [r5] gates of the multi-step quantities are set from data: the reference moves 6e-6 from ITSELF here when only its CPU thread count changes
    # (wrapper_step_spread.npz); this build sits 1.1e-4 from the 8-thread run -- a third summation order plus exact-erf / BatchNorm expression
    # differences, two AdamW(eps 1e-4) updates deep -- and is gated at 5e-4 (2e-3 through round 4)
    _close("gt step 3 pose", m.pred_pose.detach().cpu().numpy(), g["pred_pose_step3"], 5e-4, ("gt", "pose_step3"))
    sd = m.net_AutoEncoder.state_dict()                          # BatchNorm1d running statistics after the three train-mode forwards
    for k in g.files:
        if k.startswith("buf:"):
            np.testing.assert_allclose(sd[k[4:]].cpu().numpy(), g[k], atol=2e-4, rtol=1e-3, err_msg=k)


class _Acc:
    def __init__(self):
        self.rows = []

    def update(self, d):
        self.rows.append({k: float(v) for k, v in d.items()})


@pytest.mark.parametrize("tag,use_gt", [("gt", True), ("rgb", False)])
def test_wrapper_evaluate_matches_the_reference_wrapper(tmp_path, tag, use_gt):
    """test.py's flow: create_model (isTrain False) -> load_networks('best') from the reference's file names -> evaluate().
    'gt': the pure head path; 'rgb': through both estimators (the fixture ran them over this repo's ResNet-18 restatement standing in
    for torchvision: pins the wrapper's plumbing -- which net fills which channels, concat order -- not torchvision's arithmetic)."""
    from egotap_amd import models
    g = np.load(os.path.join(GOLD, "wrapper_eval_ue_b4.npz"))
    lift, pos, rot = _state_dicts()
    save_dir = tmp_path / "gold_wrapper"
    os.makedirs(save_dir)
    for name, sd in (("HeatMap", pos), ("RotHeatMap", rot), ("AutoEncoder", lift)):
        torch.save(sd, save_dir / f"best_net_{name}.pth")
    m = models.create_model(_opt(tmp_path, False, use_gt))
    m.load_networks("best")
    m.eval()
    m.set_input(_data(4, "eval"))
    acc = _Acc()
    pose, cat, _ = m.evaluate(acc)
    torch.cuda.synchronize()
    tol = 1e-4 if use_gt else 5e-4             # from RGB: 2 x 21 conv layers in fp32 in front of the head (oracle-vs-reference distance)
    np.testing.assert_allclose(pose.cpu().numpy(), g[f"{tag}_pred_pose"], atol=tol)
    np.testing.assert_allclose(cat.reshape(-1)[::997].cpu().numpy(), g[f"{tag}_heatmap_cat_sample"], atol=0 if use_gt else 3e-4)
    np.testing.assert_allclose([float(cat.double().sum()), float(cat.double().abs().sum())], g[f"{tag}_heatmap_cat_stats"], rtol=1e-9 if use_gt else 1e-4)
    assert len(acc.rows) == 4
    np.testing.assert_allclose([r["mpjpe"] for r in acc.rows], g[f"{tag}_mpjpe"], rtol=1e-4)
    np.testing.assert_allclose([r["pa_mpjpe"] for r in acc.rows], g[f"{tag}_pa_mpjpe"], rtol=1e-3)


def test_egocap_wrapper_step_and_evaluate_match_the_reference_wrapper(tmp_path):
    """the EgoCap preset (17 heatmaps per eye, 17 joints, no head joint, zero root prepended in the bone loss, utils/loss.py:56-74)
    through the reference's wrapper: first step (lr 0), third step (two updates later) and evaluate() from ground-truth heatmaps.

    Gradients at 102 encoder rows: the LeakyReLU behind every BatchNorm1d makes the gradient discontinuous where its input crosses
    zero, and fc1's 16384-long fp32 dot products leave the sign of an input within ~1e-5 of zero to the summation order (about one
    element per step; the whole row of dA behind it then differs by a few per cent of the typical magnitude, and so does every ViT
    gradient).  So the chain is: (1) tests/test_oracle_golden.py pins the float64 oracle to the reference wrapper's gradients of this
    very fixture at 1e-3 (CPU); (2) here the HIP gradients are compared with the oracle evaluated with the LeakyReLU branches the HIP
    forward took, and (3) those branches may differ from the oracle's own only where the input is ~0."""
    from egotap_amd import models, spec, training
    from egotap_amd import train_ops as T
    from egotap_amd.options import preset_defaults
    from oracle import lift_ref as O
    g = np.load(os.path.join(GOLD, "wrapper_ec.npz"))
    p = spec.lift_preset("EgoCap")
    sd_np = synth_state_dict(spec.lift_state_spec(p))
    sd_ec = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    for sub, nh, salt in (("hm_pos", 17, "hm_pos."), ("hm_sin", 34, "hm_rot.")):
        os.makedirs(tmp_path / sub)
        torch.save({k: torch.from_numpy(v) for k, v in synth_hm_state_dict(nh, salt).items()}, tmp_path / sub / "best_net_HeatMap.pth")
    opt = preset_defaults("EgoCap")
    ref = _opt(tmp_path, True, True)
    for k in ("model", "isTrain", "use_amp", "gpu_ids", "log_dir", "experiment_name", "use_gt_heatmap", "path_to_trained_heatmap", "optimizer_type", "lr", "opt_eps",
              "weight_decay", "lr_policy", "niter", "niter_decay", "epoch_iter_cnt", "epoch_count", "lambda_mpjpe", "lambda_cos_sim"):
        setattr(opt, k, getattr(ref, k))
    m = models.create_model(opt)
    m.net_AutoEncoder.load_state_dict(sd_ec, strict=True)
    assert list(m.loss_names) == list(g["loss_names"])

    def data(B, tag):
        hm = torch.from_numpy(synth_input(f"wrap_hm_{tag}", (B, 102, 64, 64)))
        return {"input_rgb_left": torch.from_numpy(synth_input(f"wrap_rgbL_{tag}", (B, 3, 256, 256), -2.0, 2.0)),
                "input_rgb_right": torch.from_numpy(synth_input(f"wrap_rgbR_{tag}", (B, 3, 256, 256), -2.0, 2.0)),
                "gt_heatmap_left": hm[:, :17], "gt_heatmap_right": hm[:, 17:34], "gt_limb_heatmap_left": hm[:, 34:68], "gt_limb_heatmap_right": hm[:, 68:],
                "gt_local_pose": torch.from_numpy(synth_input(f"wrap_gt_{tag}", (B, 17, 3), -20.0, 20.0))}
    step = data(3, "ec_step")
    m.set_input(step)
    # the operator-by-operator composition of the step (bit-identical to the one-call ABI, tests/test_gpu_train_step.py) hands every
    # block's activation to Python: record which LeakyReLU branch each element took
    taken, real = [], T.bn_lrelu_bwd

    def spy(z, y, *a, **k):
        taken.append((y > 0).cpu())
        return real(z, y, *a, **k)
    m.net_AutoEncoder.one_call_training = False
    training.T.bn_lrelu_bwd = spy
    try:
        m.optimize_parameters()
        torch.cuda.synchronize()
    finally:
        training.T.bn_lrelu_bwd = real
        m.net_AutoEncoder.one_call_training = True
    e1 = m.get_current_errors()
    assert list(e1.keys()) == list(g["errors_keys"])
    np.testing.assert_allclose([e1[k] for k in e1], g["errors_step1"], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(m.pred_pose.detach().cpu().numpy(), g["pred_pose_step1"], atol=1e-4)
    params = dict(m.net_AutoEncoder.named_parameters())
    assert sorted(k for k, v in params.items() if v.grad is not None) == sorted(g["grad_keys"])
    order = [f"{e}_heatmap_encoder.fc{j}" for e in ("rot", "pos") for j in (3, 2, 1)]          # the backward's order
    assert len(taken) == 6
    hook = {"masks": dict(zip(order, taken)), "pre": {}}
    hm = torch.cat([step["gt_heatmap_left"], step["gt_heatmap_right"], step["gt_limb_heatmap_left"], step["gt_limb_heatmap_right"]], 1)
    want = O.train_step(hm.double(), step["gt_local_pose"].double(), O.to_torch_sd(sd_np, torch.float64), p, lrelu=hook)["grads"]
    flips = 0
    for k in order:
        differ = hook["masks"][k] != (hook["pre"][k] > 0)
        flips += int(differ.sum())
        assert float(hook["pre"][k][differ].abs().max()) < 1e-4 if differ.any() else True, k          # BatchNorm outputs are O(1)
    print(f"LeakyReLU branches that differ from the float64 oracle's: {flips} of {sum(t.numel() for t in taken)}")
    norms = dict(zip(g["grad_keys"], g["grad_norms"]))
    for k in g["grad_keys"]:
        gr = params[k].grad
        scale = max(norms[k] / np.sqrt(gr.numel()), 1e-12)
        err = float((gr.double().cpu() - want[k]).abs().max())
        assert err <= 5e-3 * scale + 1e-8, f"{k}: err {err:.3e} vs typical magnitude {scale:.3e}"
        if flips == 0:      # same branches everywhere: the reference wrapper's own numbers apply directly
            assert np.abs(_strided(gr) - g["g:" + k]).max() <= 5e-3 * scale + 1e-8, k
    m.update_learning_rate()
    m.optimize_parameters()
    m.update_learning_rate()
    m.optimize_parameters()
    torch.cuda.synchronize()
    e3 = m.get_current_errors()
    np.testing.assert_allclose([e3[k] for k in e3], g["errors_step3"], rtol=2e-3, atol=2e-5)
    _close("EgoCap step 3 pose", m.pred_pose.detach().cpu().numpy(), g["pred_pose_step3"], 1e-3)
    m.eval()
    m.set_input(data(4, "ec_eval"))
    acc = _Acc()
    pose, _, _ = m.evaluate(acc)
    torch.cuda.synchronize()
    _close("EgoCap eval pose after three steps", pose.cpu().numpy(), g["eval_pred_pose"], 2e-3)          # three AdamW steps away from the initial weights
    np.testing.assert_allclose([r["mpjpe"] for r in acc.rows], g["eval_mpjpe"], rtol=1e-3)
    np.testing.assert_allclose([r["pa_mpjpe"] for r in acc.rows], g["eval_pa_mpjpe"], rtol=2e-3)


def test_wrapper_steps_from_rgb_match_the_reference_wrapper(tmp_path):
    """The path train.py runs WITHOUT --use_gt_heatmap (tools/make_golden.py gen_wrapper_rgb, the reference's own wrapper): frozen
    estimators loaded from <dir>_pos / <dir>_sin, model.train() (train.py:91) -> their BatchNorm2d normalises with BATCH statistics,
    per eye, and its running statistics keep drifting (egotap_autoencoder_model.py:127-129 freezes parameters only); three
    optimize_parameters() from RGB; validation as utils/evaluate.py:149-168 does it (model.eval(), evaluate() on the drifted running
    statistics, model.train()); and evaluate() straight from train mode, where set_eval_mode() (:325-327) forgets net_RotHeatMap.
    All of it is the DEFAULT behaviour of egotap_amd.models (round 4); opt.frozen_heatmap_bn_eval is the opt-out."""
    from egotap_amd import models
    g = np.load(os.path.join(GOLD, "wrapper_step_rgb_ue_b2.npz"))
    lift, pos, rot = _state_dicts()
    for sub, sd in (("hm_pos", pos), ("hm_sin", rot)):
        os.makedirs(tmp_path / sub)
        torch.save(sd, tmp_path / sub / "best_net_HeatMap.pth")
    m = models.create_model(_opt(tmp_path, True, False))
    m.net_AutoEncoder.load_state_dict(lift, strict=True)
    m.train()                                                    # train.py:91
    assert [int(n.training) for n in (m.net_HeatMap, m.net_RotHeatMap, m.net_AutoEncoder)] == list(g["modes_after_train"])
    frozen0 = {tag: {k: v.clone() for k, v in net.named_parameters()} for tag, net in (("pos", m.net_HeatMap), ("rot", m.net_RotHeatMap))}
    m.set_input(_data(2, "rgbstep"))
    norms = dict(zip(g["grad_keys"], g["grad_norms"]))
    for step in (1, 2, 3):
        m.optimize_parameters()
        torch.cuda.synchronize()
        errs = m.get_current_errors()
        assert list(errs.keys()) == list(g["errors_keys"])
        cat = m.pred_heatmap_cat
        # heatmaps: 2 x 21 convolutions with batch-statistics BatchNorm (B = 2: 128 values per channel in layer4) in fp32
        np.testing.assert_allclose(cat.reshape(-1)[::997].cpu().numpy(), g[f"cat_sample_step{step}"], atol=5e-4, err_msg=f"step {step}")
        np.testing.assert_allclose([float(cat.double().sum()), float(cat.double().abs().sum())], g[f"cat_stats_step{step}"], rtol=2e-4)
        tol = 5e-4                                               # [r5] step 3 too (2e-3 before): observed 3e-5, the reference's own thread-count spread is 1e-5
        np.testing.assert_allclose([errs[k] for k in errs], g[f"errors_step{step}"], rtol=max(tol, 1e-3), atol=2e-5)
        _close(f"rgb step {step} pose", m.pred_pose.detach().cpu().numpy(), g[f"pred_pose_step{step}"], tol, ("rgb", f"pose_step{step}"))
        if step == 1:
            assert not cat.requires_grad and int(g["cat_requires_grad"][0]) == 0
            params = dict(m.net_AutoEncoder.named_parameters())
            assert sorted(k for k, v in params.items() if v.grad is not None) == sorted(g["grad_keys"])
            bad = []
            for k in g["grad_keys"]:
                gr = params[k].grad
                scale = max(norms[k] / np.sqrt(gr.numel()), 1e-12)
                err = np.abs(_strided(gr) - g["g:" + k]).max()
                if err > 2e-2 * scale + 1e-8:                    # (a LeakyReLU input within 1e-5 of zero may take the other branch: DESIGN section 5)
                    bad.append((k, err, scale))
            assert len(bad) == 0, bad
        m.update_learning_rate()
    # the estimators' BatchNorm2d running statistics after 3 steps x 2 eyes, every layer of both nets; parameters untouched
    for tag, net in (("pos", m.net_HeatMap), ("rot", m.net_RotHeatMap)):
        sd = net.state_dict()
        n = 0
        for k in g.files:
            if not k.startswith(f"buf_{tag}:"):
                continue
            key = k.split(":", 1)[1]
            want = g[k]
            if key.endswith("num_batches_tracked"):
                assert int(sd[key]) == int(want) == 6, key
            else:
                np.testing.assert_allclose(sd[key].cpu().numpy(), want, rtol=2e-3, atol=2e-4 * max(1.0, float(np.abs(want).max())), err_msg=key)
            n += 1
        assert n == 60                                           # 20 BatchNorm2d layers x (mean, var, count)
        for k, v in net.named_parameters():
            assert torch.equal(v, frozen0[tag][k]) and not v.requires_grad, k
        np.testing.assert_allclose(float(sum(p.double().sum() for p in net.parameters())), g[f"frozen_param_sum_{tag}"][0], rtol=1e-9)

    # validation between epochs: model.eval() -> evaluate() on the drifted running statistics -> model.train()
    m.eval()
    m.set_input(_data(4, "rgbeval"))
    acc = _Acc()
    pose, cat, _ = m.evaluate(acc)
    torch.cuda.synchronize()
    np.testing.assert_allclose(cat.reshape(-1)[::997].cpu().numpy(), g["eval_cat_sample"], atol=1e-3)
    # (3e-3 stays: the reference ITSELF moves 4.2e-3 on this quantity between 8 threads and 1 -- evaluation normalises with BatchNorm1d running
    # statistics gathered over three batches of two frames; observed here: 2.5e-5)
    _close("rgb eval pose after three steps", pose.cpu().numpy(), g["eval_pred_pose"], 3e-3, ("rgb", "eval_pose"))
    np.testing.assert_allclose([r["mpjpe"] for r in acc.rows], g["eval_mpjpe"], rtol=1e-3)
    np.testing.assert_allclose([r["pa_mpjpe"] for r in acc.rows], g["eval_pa_mpjpe"], rtol=2e-3)
    m.train()
    # evaluate() straight from train mode: the limb estimator stays on batch statistics (and counts two more batches)
    acc = _Acc()
    pose, cat, _ = m.evaluate(acc)
    torch.cuda.synchronize()
    assert [int(n.training) for n in (m.net_HeatMap, m.net_RotHeatMap, m.net_AutoEncoder)] == list(g["quirk_modes_after_evaluate"])
    k = "backbone.backbone.backbone.bn1.num_batches_tracked"
    assert [int(m.net_HeatMap.state_dict()[k]), int(m.net_RotHeatMap.state_dict()[k])] == list(g["quirk_num_batches_tracked"])
    np.testing.assert_allclose(cat.reshape(-1)[::997].cpu().numpy(), g["quirk_cat_sample"], atol=1e-3)
    _close("rgb evaluate() straight from train mode, pose", pose.cpu().numpy(), g["quirk_pred_pose"], 1e-3)


def test_wrapper_evaluate_batch_of_two_prints_the_reference_pa_mpjpe(tmp_path):
    """utils/util.py:337: for a batch of 2 (or 3) frames the reference aligns the wrong axes; the reference wrapper's evaluate() on the
    first two frames of the G7 batch is in the fixture (quirk_b2_*).  Default = the reference's number; the switch restores the
    batch-independent one."""
    from egotap_amd import models
    g = np.load(os.path.join(GOLD, "wrapper_eval_ue_b4.npz"))
    lift, pos, rot = _state_dicts()
    save_dir = tmp_path / "gold_wrapper"
    os.makedirs(save_dir)
    for name, sd in (("HeatMap", pos), ("RotHeatMap", rot), ("AutoEncoder", lift)):
        torch.save(sd, save_dir / f"best_net_{name}.pth")
    m = models.create_model(_opt(tmp_path, False, True))
    m.load_networks("best")
    m.eval()
    m.set_input({k: v[:2] for k, v in _data(4, "eval").items()})
    acc = _Acc()
    m.evaluate(acc)
    np.testing.assert_allclose([r["mpjpe"] for r in acc.rows], g["quirk_b2_gt_mpjpe"], rtol=1e-4)
    np.testing.assert_allclose([r["pa_mpjpe"] for r in acc.rows], g["quirk_b2_gt_pa_mpjpe"], rtol=1e-3)
    m.opt.pa_mpjpe_reference_batch_axes = False
    acc = _Acc()
    m.evaluate(acc)
    np.testing.assert_allclose([r["pa_mpjpe"] for r in acc.rows], g["gt_pa_mpjpe"][:2], rtol=1e-3)
