#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules on CPU.

Runs only in the build container (needs /root/reference).  Nothing from the
reference is copied: its modules are imported in place, their parameters are
overwritten with the hash-RNG weights of ``egotap_amd.synthetic`` (so no weight
is ever stored), they are run on synthetic inputs and the outputs are saved as
small fixtures.  The GPU box only ever sees the fixtures.

Shims applied before the import (each fills a hole left by a package that is
missing or newer here than the reference pins; none touches reference
arithmetic) -- SURVEY.md section 8(c):
  1. transformers.pytorch_utils.find_pruneable_heads_and_indices (removed in
     transformers 5; only used by the never-called prune_heads)
  2. ViTModel.get_head_mask -> [None] * num_layers (4.33 semantics for None)
  3. sys.modules['torchvision'(.models)] -> stub whose resnet18 is this repo's
     own ResNet-18 restatement (oracle/hm_ref.py); torchvision is not installed
  4. empty skimage / skimage.draw / cv2 stubs (transitive imports, never run)
  5. torch.Tensor.cuda -> .to(device or 'cpu') (the wrapper hard-codes .cuda())

usage: python tools/make_golden.py [--only lift,pu,fc,loss,hm]
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")

from egotap_amd.synthetic import synth_tensor, synth_input  # noqa: E402


# --------------------------------------------------------------------------
def apply_shims():
    import transformers.pytorch_utils as tpu

    if not hasattr(tpu, "find_pruneable_heads_and_indices"):
        tpu.find_pruneable_heads_and_indices = lambda *a, **k: None
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tv.models = tvm
    try:
        from oracle import hm_ref

        tvm.resnet18 = lambda pretrained=False, **k: hm_ref.ResNet18Holder()
    except Exception:  # lifting-head fixtures do not need a backbone
        pass
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    for n in ("skimage", "skimage.draw", "cv2"):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["skimage.draw"].line_aa = None
    torch.Tensor.cuda = lambda self, device=None, **k: self.to(device or "cpu")
    sys.path.insert(0, REF)
    import model.modeling_vit as mv

    if not hasattr(mv.ViTModel, "get_head_mask"):
        mv.ViTModel.get_head_mask = lambda self, head_mask, n, *a, **k: [None] * n
    else:
        try:
            mv.ViTModel.get_head_mask(object.__new__(mv.ViTModel), None, 1)
        except Exception:
            mv.ViTModel.get_head_mask = lambda self, head_mask, n, *a, **k: [None] * n


def make_opt(preset: str, hm: int = 64):
    o = types.SimpleNamespace()
    o.joint_preset = preset
    nj = 15 if preset == "UnrealEgo" else 17
    o.num_heatmap = nj
    o.num_rot_heatmap = nj
    o.heatmap_type = "sin"
    o.ae_hidden_size = 128
    o.patched_heatmap_ae = True
    o.skel_layer = "PU"
    o.load_size_heatmap = [hm, hm]
    o.estimate_head = preset == "UnrealEgo"
    o.stereo = True
    o.init_ImageNet = False
    o.model_name = "resnet18"
    return o


def load_synth(module: torch.nn.Module, salt: str = ""):
    """Overwrite every parameter and buffer with its hash-RNG value."""
    sd = module.state_dict()
    new = {k: torch.from_numpy(synth_tensor(salt + k, tuple(v.shape))) for k, v in sd.items()}
    module.load_state_dict(new, strict=True)


def sample(t: torch.Tensor, stride: int = 997) -> np.ndarray:
    return t.detach().reshape(-1)[::stride].to(torch.float32).numpy().copy()


def stats(t: torch.Tensor) -> np.ndarray:
    d = t.detach().double()
    return np.array([d.sum().item(), d.abs().sum().item()], dtype=np.float64)


# --------------------------------------------------------------------------
def gen_lift(tag: str, preset: str, batch: int = 2):
    import model.net_architecture as na

    torch.manual_seed(0)
    opt = make_opt(preset)
    net = na.EgoTAPAutoEncoder(opt, input_channel_scale=2)
    load_synth(net)
    net.eval()
    nj = opt.num_heatmap
    hm = torch.from_numpy(synth_input(f"hm_{tag}", (batch, 6 * nj, 64, 64)))

    cap = {}

    def hook(name):
        def f(mod, inp, out):
            cap[name] = out[0] if isinstance(out, tuple) else out
        return f

    vit = net.pos_heatmap_encoder.vit
    hs = [vit.embeddings.register_forward_hook(hook("emb"))]
    for i, layer in enumerate(vit.encoder.layer):
        hs.append(layer.register_forward_hook(hook(f"layer{i}")))
    hs.append(vit.layernorm.register_forward_hook(hook("final_ln")))
    hs.append(net.pos_heatmap_encoder.register_forward_hook(hook("pos_embed")))
    hs.append(net.rot_heatmap_encoder.register_forward_hook(hook("rot_embed")))
    with torch.no_grad():
        pose, rot, indep, out_hm = net(hm)
    for h in hs:
        h.remove()
    out = {
        "preset": np.array(preset),
        "batch": np.array(batch),
        "pose": pose.detach().numpy(),
        "pos_embed": cap["pos_embed"].detach().numpy(),
        "rot_embed": cap["rot_embed"].detach().numpy(),
        "skel_embed": net.skel_embed.detach().numpy(),
        "rot_is_zero": np.array(float(rot.abs().max()) == 0.0 and tuple(rot.shape) == (batch, 3 * nj)),
        "indep_is_zero": np.array(float(indep.abs().max()) == 0.0 and tuple(indep.shape) == (batch, 6 * nj)),
        "out_hm_is_zero": np.array(float(out_hm.abs().max()) == 0.0 and tuple(out_hm.shape) == tuple(hm.shape)),
    }
    ref_sd = net.state_dict()
    out["state_keys"] = np.array(list(ref_sd.keys()))
    out["state_shapes"] = np.array(["x".join(str(d) for d in v.shape) for v in ref_sd.values()])
    out["param_keys"] = np.array([k for k, _ in net.named_parameters()])
    for k in ("emb", "layer0", "layer1", "layer2", "final_ln"):
        out[k + "_sample"] = sample(cap[k])
        out[k + "_stats"] = stats(cap[k])
    np.savez_compressed(os.path.join(GOLD, f"lift_fwd_{tag}_b{batch}.npz"), **out)
    print(f"lift_fwd_{tag}: pose[0,:2] =", pose[0, :2].tolist())


def gen_pu():
    import model.net_architecture as na

    for tag, preset in (("ue", "UnrealEgo"), ("ec", "EgoCap")):
        opt = make_opt(preset)
        skel = na.SkelNet(opt, input_size=256, bridge_size=256, num_layers=2, batch_first=False, layer_type="PU")
        load_synth(skel, salt="skel_sequential_layer.")
        skel.eval()
        nj = opt.num_heatmap
        x = torch.from_numpy(synth_input(f"pu_x_{tag}", (nj, 3, 256), -1.0, 1.0))
        b = torch.from_numpy(synth_input(f"pu_b_{tag}", (nj, 3, 256), -1.0, 1.0))
        with torch.no_grad():
            y = skel(input=x, bridge=b)
        np.savez_compressed(os.path.join(GOLD, f"pu_chain_{tag}.npz"), out=y.numpy())
        print(f"pu_chain_{tag}:", tuple(y.shape), float(y.abs().mean()))


def gen_fcblock():
    import model.network_utils as nu

    blk = nu.make_fc_layer(96, 64)
    load_synth(blk, salt="fcblock.")
    x = torch.from_numpy(synth_input("fcblock_x", (24, 96), -1.0, 1.0))
    blk.eval()
    with torch.no_grad():
        y_eval = blk(x)
    blk.train()
    with torch.no_grad():
        y_train = blk(x)
    np.savez_compressed(
        os.path.join(GOLD, "fcblock.npz"),
        y_eval=y_eval.numpy(),
        y_train=y_train.numpy(),
        running_mean=blk.bn.running_mean.numpy(),
        running_var=blk.bn.running_var.numpy(),
        num_batches_tracked=blk.bn.num_batches_tracked.numpy(),
    )
    print("fcblock ok")


def gen_loss():
    import utils.loss as L

    for tag, preset, nj, head in (("ue", "UnrealEgo", 16, True), ("ec", "EgoCap", 17, False)):
        pred = torch.from_numpy(synth_input(f"loss_pred_{tag}", (5, nj, 3), -20.0, 20.0)).requires_grad_(True)
        gt = torch.from_numpy(synth_input(f"loss_gt_{tag}", (5, nj, 3), -20.0, 20.0))
        mp = L.LossFuncMPJPE()(pred, gt)
        cs = L.LossFuncCosSim(joint_preset=preset, estimate_head=head)(pred, gt)
        total = 0.1 * mp + (-0.01) * 0.1 * cs
        (g,) = torch.autograd.grad(total, pred)
        np.savez_compressed(
            os.path.join(GOLD, f"loss_{tag}.npz"),
            mpjpe=mp.detach().numpy(), cos_sim=cs.detach().numpy(), dtotal_dpred=g.numpy(),
        )
        print(f"loss_{tag}:", float(mp), float(cs))


def gen_train():
    """One training forward/backward of the reference lifting head (train-mode BatchNorm, MPJPE + cos-sim loss of
    egotap_autoencoder_model.py:284-296) and one AdamW step (lr 1e-3, eps 1e-4, wd 0): gradients and updated values."""
    import model.net_architecture as na
    import utils.loss as L

    opt = make_opt("UnrealEgo")
    net = na.EgoTAPAutoEncoder(opt, input_channel_scale=2)
    load_synth(net)
    net.train()
    B = 2
    hm = torch.from_numpy(synth_input("hm_train", (B, 90, 64, 64)))
    gt = torch.from_numpy(synth_input("gt_train", (B, 16, 3), -1.0, 1.0))
    optim = torch.optim.AdamW(net.parameters(), lr=1e-3, eps=1e-4, weight_decay=0.0)
    optim.zero_grad()
    pose = net(hm)[0]
    lam_m, lam_c = 0.1, -0.01
    loss_pose = L.LossFuncMPJPE()(pose, gt) * lam_m
    loss_cos = L.LossFuncCosSim(joint_preset="UnrealEgo", estimate_head=True)(pose, gt) * lam_c * lam_m
    (loss_pose + loss_cos).backward()
    out = {"pose": pose.detach().numpy(), "loss_pose": loss_pose.detach().numpy(), "loss_cos_sim": loss_cos.detach().numpy()}
    names, norms, no_grad = [], [], []
    for k, prm in net.named_parameters():
        if prm.grad is None:
            no_grad.append(k)
            continue
        names.append(k)
        norms.append(float(prm.grad.double().norm()))
        out["g:" + k] = prm.grad.reshape(-1)[:: max(1, prm.numel() // 257)].numpy().copy()
    out["grad_keys"] = np.array(names)
    out["grad_norms"] = np.array(norms)
    out["no_grad_keys"] = np.array(no_grad)
    for k, v in net.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"):
            out["buf:" + k] = v.numpy().copy()
    optim.step()
    for k, prm in net.named_parameters():
        if prm.grad is not None:
            out["p:" + k] = prm.detach().reshape(-1)[:: max(1, prm.numel() // 257)].numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "train_step_ue_b2.npz"), **out)
    print("train_step: losses", float(loss_pose), float(loss_cos), "params without grad:", no_grad)


def _wrapper_opt(tmp, is_train, use_gt_heatmap, preset="UnrealEgo"):
    """the shipped PoseEstimator flag set (scripts/train/PoseEstimator/unrealego.sh, scripts/test/unrealego.sh; egocap.sh for the
    EgoCap preset) without --use_amp (fp16 autocast needs a GPU) on CPU"""
    opt = make_opt(preset)
    opt.model, opt.isTrain, opt.use_amp, opt.gpu_ids, opt.distributed = "egotap_autoencoder", is_train, False, [], False
    opt.log_dir, opt.experiment_name, opt.init_type = tmp, "gold_wrapper", "kaiming"
    opt.use_gt_heatmap = use_gt_heatmap
    opt.path_to_trained_heatmap = "hm/best_net_HeatMap.pth" if is_train else None      # -> hm_pos/ and hm_sin/ (egotap_autoencoder_model.py:113-126)
    opt.optimizer_type, opt.lr, opt.opt_eps, opt.weight_decay, opt.lr_policy = "AdamW", 1e-3, 1e-4, 0.0, "cos_anneal_warmup"
    opt.niter, opt.niter_decay, opt.epoch_iter_cnt, opt.epoch_count = 1, 15, 4, 1
    opt.lambda_mpjpe, opt.lambda_cos_sim, opt.lambda_heatmap, opt.lambda_rot_heatmap = 0.1, -0.01, 1.0, 1.0
    return opt


def _wrapper_data(B, tag, J=15, out_joints=16):
    hm = torch.from_numpy(synth_input(f"wrap_hm_{tag}", (B, 6 * J, 64, 64)))
    return {
        "input_rgb_left": torch.from_numpy(synth_input(f"wrap_rgbL_{tag}", (B, 3, 256, 256), -2.0, 2.0)),
        "input_rgb_right": torch.from_numpy(synth_input(f"wrap_rgbR_{tag}", (B, 3, 256, 256), -2.0, 2.0)),
        "gt_heatmap_left": hm[:, :J], "gt_heatmap_right": hm[:, J:2 * J],
        "gt_limb_heatmap_left": hm[:, 2 * J:4 * J], "gt_limb_heatmap_right": hm[:, 4 * J:],
        "gt_local_pose": torch.from_numpy(synth_input(f"wrap_gt_{tag}", (B, out_joints, 3), -20.0, 20.0)),
        "gt_local_rot": torch.zeros(B, out_joints, 3), "gt_limb_theta": torch.zeros(B, J),
        "gt_pelvis_left": torch.zeros(B, 3), "gt_pelvis_right": torch.zeros(B, 3),
        "gt_plength_left": torch.ones(B, 2 * J), "gt_plength_right": torch.ones(B, 2 * J),
    }


def gen_wrapper():
    """the reference's own WRAPPER (model/egotap_autoencoder_model.py), not the bare network.
    G6 wrapper_step_ue_b2: EgoTAPAutoEncoderModel built with the shipped training flags (frozen estimators loaded from
       <dir>_pos / <dir>_sin checkpoints, --use_gt_heatmap), set_input -> optimize_parameters() twice + update_learning_rate():
       get_current_errors() of both steps, every gradient of step 1 (strided sample + norm), parameters after each step, lr.
    G7 wrapper_eval_ue_b4: the test-mode wrapper, load_networks('best') from the reference's file names, model.eval(),
       evaluate(): pred_pose, pred_heatmap_cat, per-sample mpjpe / pa_mpjpe -- once from ground-truth heatmaps (pure head path) and
       once from RGB through the two estimators (over this repo's ResNet-18 stand-in: pins the wrapper's estimator plumbing, channel
       order and concat, not torchvision's arithmetic)."""
    import tempfile
    from model.egotap_autoencoder_model import EgoTAPAutoEncoderModel
    from egotap_amd import spec
    from egotap_amd.synthetic import synth_hm_state_dict, synth_state_dict

    tmp = tempfile.mkdtemp(prefix="egotap_gold_")
    sd_lift = {k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(spec.lift_preset("UnrealEgo"))).items()}
    sd_pos = {k: torch.from_numpy(v) for k, v in synth_hm_state_dict(15, "hm_pos.").items()}
    sd_rot = {k: torch.from_numpy(v) for k, v in synth_hm_state_dict(30, "hm_rot.").items()}
    for sub, sd in (("hm_pos", sd_pos), ("hm_sin", sd_rot)):
        os.makedirs(os.path.join(tmp, sub))
        torch.save(sd, os.path.join(tmp, sub, "best_net_HeatMap.pth"))

    # ---- G6: two optimisation steps of the training wrapper
    opt = _wrapper_opt(tmp, True, True)
    m = EgoTAPAutoEncoderModel()
    m.initialize(opt)
    m.net_AutoEncoder.load_state_dict(sd_lift, strict=True)
    m.train()                                                   # train.py:91
    B = 2
    m.set_input(_wrapper_data(B, "step"))
    out = {"loss_names": np.array(m.loss_names), "lr_before": np.array([m.optimizers[0].param_groups[0]["lr"]])}
    m.optimize_parameters()
    errs = m.get_current_errors()
    out["errors_keys"] = np.array(list(errs.keys()))
    out["errors_step1"] = np.array([errs[k] for k in errs], dtype=np.float64)
    out["pred_pose_step1"] = m.pred_pose.detach().numpy().copy()
    out["pred_heatmap_cat_stats"] = stats(m.pred_heatmap_cat)
    out["pred_heatmap_cat_sample"] = sample(m.pred_heatmap_cat, 997)
    out["rec_shapes"] = np.array([list(getattr(m, n).shape) for n in ("pred_heatmap_left_rec", "pred_heatmap_right_rec",
                                                                      "pred_limb_heatmap_left_rec", "pred_limb_heatmap_right_rec")])
    out["rec_abs_sum"] = np.array([float(m.pred_heatmap_rec_cat.abs().sum())])
    out["pred_rot_shape"], out["pred_indep_pos_shape"] = np.array(m.pred_rot.shape), np.array(m.pred_indep_pos.shape)
    names, norms, no_grad = [], [], []
    for k, prm in m.net_AutoEncoder.named_parameters():
        if prm.grad is None:
            no_grad.append(k)
            continue
        names.append(k)
        norms.append(float(prm.grad.double().norm()))
        out["g:" + k] = prm.grad.reshape(-1)[:: max(1, prm.numel() // 257)].numpy().copy()
        out["p1:" + k] = prm.detach().reshape(-1)[:: max(1, prm.numel() // 257)].numpy().copy()
    out["grad_keys"], out["grad_norms"], out["no_grad_keys"] = np.array(names), np.array(norms), np.array(no_grad)
    out["frozen_requires_grad"] = np.array([int(any(q.requires_grad for q in n.parameters())) for n in (m.net_HeatMap, m.net_RotHeatMap)])
    m.update_learning_rate()
    out["lr_after_step1"] = np.array([m.optimizers[0].param_groups[0]["lr"]])
    m.optimize_parameters()
    errs2 = m.get_current_errors()
    out["errors_step2"] = np.array([errs2[k] for k in errs2], dtype=np.float64)
    out["pred_pose_step2"] = m.pred_pose.detach().numpy().copy()
    chk = []
    for k in names:
        prm = dict(m.net_AutoEncoder.named_parameters())[k]
        out["p2:" + k] = prm.detach().reshape(-1)[:: max(1, prm.numel() // 257)].numpy().copy()
        chk.append(float(prm.detach().double().sum()))
    out["param_sums_step2"] = np.array(chk)
    m.update_learning_rate()
    m.optimize_parameters()                                     # the first step ran at lr 0 (warm-up from zero): step 3 sees moved parameters
    errs3 = m.get_current_errors()
    out["errors_step3"] = np.array([errs3[k] for k in errs3], dtype=np.float64)
    out["pred_pose_step3"] = m.pred_pose.detach().numpy().copy()
    out["lr_after_step2"] = np.array([m.optimizers[0].param_groups[0]["lr"]])
    for k, v in m.net_AutoEncoder.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"):
            out["buf:" + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "wrapper_step_ue_b2.npz"), **out)
    print("wrapper_step:", errs, errs2, errs3, "no grad:", no_grad, "lr", out["lr_before"], out["lr_after_step1"])

    # ---- G7: the test-mode wrapper
    save_dir = os.path.join(tmp, "gold_wrapper")
    os.makedirs(save_dir, exist_ok=True)
    for name, sd in (("HeatMap", sd_pos), ("RotHeatMap", sd_rot), ("AutoEncoder", sd_lift)):
        torch.save(sd, os.path.join(save_dir, f"best_net_{name}.pth"))

    class Acc:                                                  # utils/evaluate.py's RunningAverageDict as far as evaluate() uses it
        def __init__(self):
            self.rows = []

        def update(self, d):
            self.rows.append({k: float(v.detach()) for k, v in d.items()})

    # B = 4, not 2: utils/util.py:337 skips its transpose when S1.shape[0] is 2 or 3 (meant for unbatched 3 x N / 2 x N input), so for a
    # BATCH of 2 or 3 frames the reference aligns the wrong axes (a 16 x 16 "rotation" over 3 "points").  That accident is not
    # reproduced by this repo; the B = 2 values are recorded below only to document it.
    out = {}
    BE = 4
    for tag, use_gt in (("gt", True), ("rgb", False)):
        opt = _wrapper_opt(tmp, False, use_gt)
        m = EgoTAPAutoEncoderModel()
        m.initialize(opt)
        m.load_networks("best")                                 # test.py:27
        m.eval()                                                # utils/evaluate.py:93
        m.set_input(_wrapper_data(BE, "eval"))
        acc = Acc()
        pose, cat, _ = m.evaluate(acc)
        out[f"{tag}_pred_pose"] = pose.detach().numpy().copy()      # (net_architecture.py:683 re-enables grad inside the head)
        out[f"{tag}_heatmap_cat_sample"] = sample(cat, 997)
        out[f"{tag}_heatmap_cat_stats"] = stats(cat)
        out[f"{tag}_mpjpe"] = np.array([r["mpjpe"] for r in acc.rows])
        out[f"{tag}_pa_mpjpe"] = np.array([r["pa_mpjpe"] for r in acc.rows])
        print(f"wrapper_eval[{tag}]: mpjpe", out[f"{tag}_mpjpe"], "pa", out[f"{tag}_pa_mpjpe"])
    data2 = {k: v[:2] for k, v in _wrapper_data(BE, "eval").items()}
    m.opt.use_gt_heatmap = True
    m.set_input(data2)
    acc = Acc()
    m.evaluate(acc)
    out["quirk_b2_gt_pa_mpjpe"] = np.array([r["pa_mpjpe"] for r in acc.rows])
    out["quirk_b2_gt_mpjpe"] = np.array([r["mpjpe"] for r in acc.rows])
    print("batch-of-2 quirk: pa_mpjpe", out["quirk_b2_gt_pa_mpjpe"], "vs the same frames in the batch of 4:", out["gt_pa_mpjpe"][:2])
    np.savez_compressed(os.path.join(GOLD, "wrapper_eval_ue_b4.npz"), **out)

    # ---- the EgoCap preset through the same wrapper (17 heatmaps per eye, 17 joints, no head joint, zero root in the bone loss):
    # one optimisation step after the lr-0 warm-up step, and evaluate() from ground-truth heatmaps
    sd_ec = {k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(spec.lift_preset("EgoCap"))).items()}
    for sub, nh in (("hm_pos", 17), ("hm_sin", 34)):
        torch.save({k: torch.from_numpy(v) for k, v in synth_hm_state_dict(nh, "hm_pos." if sub == "hm_pos" else "hm_rot.").items()},
                   os.path.join(tmp, sub, "best_net_HeatMap.pth"))
    opt = _wrapper_opt(tmp, True, True, "EgoCap")
    m = EgoTAPAutoEncoderModel()
    m.initialize(opt)
    m.net_AutoEncoder.load_state_dict(sd_ec, strict=True)
    m.train()
    m.set_input(_wrapper_data(3, "ec_step", 17, 17))
    m.optimize_parameters()
    out = {"loss_names": np.array(m.loss_names)}
    e1 = m.get_current_errors()
    out["errors_keys"], out["errors_step1"] = np.array(list(e1.keys())), np.array([e1[k] for k in e1], dtype=np.float64)
    out["pred_pose_step1"] = m.pred_pose.detach().numpy().copy()
    names, norms = [], []
    for k, prm in m.net_AutoEncoder.named_parameters():
        if prm.grad is None:
            continue
        names.append(k)
        norms.append(float(prm.grad.double().norm()))
        out["g:" + k] = prm.grad.reshape(-1)[:: max(1, prm.numel() // 257)].numpy().copy()
    out["grad_keys"], out["grad_norms"] = np.array(names), np.array(norms)
    m.update_learning_rate()
    m.optimize_parameters()
    m.update_learning_rate()
    m.optimize_parameters()
    e3 = m.get_current_errors()
    out["errors_step3"] = np.array([e3[k] for k in e3], dtype=np.float64)
    out["pred_pose_step3"] = m.pred_pose.detach().numpy().copy()
    m.eval()
    m.opt.use_gt_heatmap = True
    m.set_input(_wrapper_data(4, "ec_eval", 17, 17))
    acc = Acc()
    pose, cat, _ = m.evaluate(acc)
    out["eval_pred_pose"] = pose.detach().numpy().copy()
    out["eval_mpjpe"] = np.array([r["mpjpe"] for r in acc.rows])
    out["eval_pa_mpjpe"] = np.array([r["pa_mpjpe"] for r in acc.rows])
    np.savez_compressed(os.path.join(GOLD, "wrapper_ec.npz"), **out)
    print("wrapper_ec:", e1, e3, "eval mpjpe", out["eval_mpjpe"])
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


def gen_wrapper_rgb():
    """G6-rgb wrapper_step_rgb_ue_b2: the path train.py runs WITHOUT --use_gt_heatmap.  The reference's wrapper with the shipped
    training flags, frozen estimators from <dir>_pos / <dir>_sin, model.train() (train.py:91: the frozen estimators' BatchNorm2d is
    in batch-statistics mode and its running statistics keep drifting, egotap_autoencoder_model.py:127-129 freezes parameters only),
    three optimize_parameters() from RGB.  Recorded: pred_heatmap_cat (sample + sums) of steps 1 and 3, losses by key, poses, every
    head gradient of step 1, EVERY BatchNorm buffer of both estimators after the three steps, then -- as utils/evaluate.py:149-168
    does between epochs -- model.eval() + evaluate() from RGB on the drifted running statistics, and evaluate() WITHOUT model.eval()
    (set_eval_mode, egotap_autoencoder_model.py:325-327, forgets net_RotHeatMap: it stays on batch statistics).
    The backbone is this repo's ResNet-18 restatement standing in for torchvision, as in G7-rgb."""
    import tempfile
    from model.egotap_autoencoder_model import EgoTAPAutoEncoderModel
    from egotap_amd import spec
    from egotap_amd.synthetic import synth_hm_state_dict, synth_state_dict

    tmp = tempfile.mkdtemp(prefix="egotap_gold_rgb_")
    sd_lift = {k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(spec.lift_preset("UnrealEgo"))).items()}
    for sub, nh, salt in (("hm_pos", 15, "hm_pos."), ("hm_sin", 30, "hm_rot.")):
        os.makedirs(os.path.join(tmp, sub))
        torch.save({k: torch.from_numpy(v) for k, v in synth_hm_state_dict(nh, salt).items()}, os.path.join(tmp, sub, "best_net_HeatMap.pth"))
    opt = _wrapper_opt(tmp, True, False)
    m = EgoTAPAutoEncoderModel()
    m.initialize(opt)
    m.net_AutoEncoder.load_state_dict(sd_lift, strict=True)
    m.train()                                                   # train.py:91
    out = {"modes_after_train": np.array([int(n.training) for n in (m.net_HeatMap, m.net_RotHeatMap, m.net_AutoEncoder)])}
    B = 2
    m.set_input(_wrapper_data(B, "rgbstep"))
    for step in (1, 2, 3):
        m.optimize_parameters()
        errs = m.get_current_errors()
        out["errors_keys"] = np.array(list(errs.keys()))
        out[f"errors_step{step}"] = np.array([errs[k] for k in errs], dtype=np.float64)
        out[f"pred_pose_step{step}"] = m.pred_pose.detach().numpy().copy()
        out[f"cat_stats_step{step}"] = stats(m.pred_heatmap_cat)
        out[f"cat_sample_step{step}"] = sample(m.pred_heatmap_cat, 997)
        if step == 1:
            names, norms = [], []
            for k, prm in m.net_AutoEncoder.named_parameters():
                if prm.grad is None:
                    continue
                names.append(k)
                norms.append(float(prm.grad.double().norm()))
                out["g:" + k] = prm.grad.reshape(-1)[:: max(1, prm.numel() // 257)].numpy().copy()
            out["grad_keys"], out["grad_norms"] = np.array(names), np.array(norms)
            out["cat_requires_grad"] = np.array([int(m.pred_heatmap_cat.requires_grad)])
        m.update_learning_rate()
    for tag, net in (("pos", m.net_HeatMap), ("rot", m.net_RotHeatMap)):
        for k, v in net.state_dict().items():
            if k.startswith("backbone.backbone.backbone.") and (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")):
                out[f"buf_{tag}:" + k] = v.numpy().copy()
        out[f"frozen_param_sum_{tag}"] = np.array([float(sum(p.double().sum() for p in net.parameters()))])

    class Acc:
        def __init__(self):
            self.rows = []

        def update(self, d):
            self.rows.append({k: float(v.detach()) for k, v in d.items()})

    # validation between epochs (utils/evaluate.py:149-168): model.eval(), evaluate() per batch -- on the DRIFTED running statistics
    m.eval()
    m.set_input(_wrapper_data(4, "rgbeval"))
    acc = Acc()
    pose, cat, _ = m.evaluate(acc)
    out["eval_pred_pose"], out["eval_cat_sample"], out["eval_cat_stats"] = pose.detach().numpy().copy(), sample(cat, 997), stats(cat)
    out["eval_mpjpe"], out["eval_pa_mpjpe"] = np.array([r["mpjpe"] for r in acc.rows]), np.array([r["pa_mpjpe"] for r in acc.rows])
    m.train()                                                   # utils/evaluate.py:168
    # evaluate() straight from train mode: set_eval_mode() switches net_AutoEncoder and net_HeatMap only -> the limb estimator
    # normalises with batch statistics (and moves its running statistics once more); the position estimator uses running statistics
    acc = Acc()
    pose, cat, _ = m.evaluate(acc)
    out["quirk_modes_after_evaluate"] = np.array([int(n.training) for n in (m.net_HeatMap, m.net_RotHeatMap, m.net_AutoEncoder)])
    out["quirk_pred_pose"], out["quirk_cat_sample"], out["quirk_cat_stats"] = pose.detach().numpy().copy(), sample(cat, 997), stats(cat)
    k = "backbone.backbone.backbone.bn1.num_batches_tracked"
    out["quirk_num_batches_tracked"] = np.array([int(m.net_HeatMap.state_dict()[k]), int(m.net_RotHeatMap.state_dict()[k])])
    np.savez_compressed(os.path.join(GOLD, "wrapper_step_rgb_ue_b2.npz"), **out)
    print("wrapper_step_rgb:", {s: out[f"errors_step{s}"] for s in (1, 2, 3)}, "modes", out["modes_after_train"], out["quirk_modes_after_evaluate"],
          "tracked", out["quirk_num_batches_tracked"], "eval mpjpe", out["eval_mpjpe"])
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


def gen_wrapper_spread():
    """[r5] How far does the REFERENCE move from itself when only the summation order of its CPU kernels changes?  The reference wrapper's
    three optimize_parameters() + evaluate() (exactly gen_wrapper's / gen_wrapper_rgb's runs: same flags, weights, data) repeated with
    torch.set_num_threads(1) and (3) against the fixtures' 8 threads: MKL / oneDNN split their reductions by thread count, nothing else
    differs.  wrapper_step_spread.npz holds, per step, max |pose_T - pose_8|, the loss differences and max |heatmap_T - heatmap_8| -- the
    yardstick the multi-step gates of tests/test_gpu_wrapper_golden.py are set from (a GPU summation order is one more such order)."""
    import tempfile
    from model.egotap_autoencoder_model import EgoTAPAutoEncoderModel
    from egotap_amd import spec
    from egotap_amd.synthetic import synth_hm_state_dict, synth_state_dict

    tmp = tempfile.mkdtemp(prefix="egotap_gold_spread_")
    sd_lift = {k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(spec.lift_preset("UnrealEgo"))).items()}
    for sub, nh, salt in (("hm_pos", 15, "hm_pos."), ("hm_sin", 30, "hm_rot.")):
        os.makedirs(os.path.join(tmp, sub))
        torch.save({k: torch.from_numpy(v) for k, v in synth_hm_state_dict(nh, salt).items()}, os.path.join(tmp, sub, "best_net_HeatMap.pth"))

    class Acc:
        def __init__(self):
            self.rows = []

        def update(self, d):
            self.rows.append({k: float(v.detach()) for k, v in d.items()})

    def run(threads, use_gt):
        torch.set_num_threads(threads)
        m = EgoTAPAutoEncoderModel()
        m.initialize(_wrapper_opt(tmp, True, use_gt))
        m.net_AutoEncoder.load_state_dict(sd_lift, strict=True)
        m.train()
        m.set_input(_wrapper_data(2, "step" if use_gt else "rgbstep"))
        rec = {}
        for step in (1, 2, 3):
            m.optimize_parameters()
            errs = m.get_current_errors()
            rec[f"pose{step}"] = m.pred_pose.detach().numpy().copy()
            rec[f"errs{step}"] = np.array([errs[k] for k in errs], dtype=np.float64)
            rec[f"cat{step}"] = m.pred_heatmap_cat.detach().numpy().copy()
            m.update_learning_rate()
        m.eval()
        m.set_input(_wrapper_data(4, "eval" if use_gt else "rgbeval"))
        pose, cat, _ = m.evaluate(Acc())
        rec["eval_pose"] = pose.detach().numpy().copy()
        return rec

    out = {}
    for tag, use_gt in (("gt", True), ("rgb", False)):
        base = run(8, use_gt)
        for threads in (1, 3):
            other = run(threads, use_gt)
            for step in (1, 2, 3):
                out[f"{tag}_t{threads}_pose_step{step}"] = np.array([np.abs(other[f"pose{step}"] - base[f"pose{step}"]).max()])
                out[f"{tag}_t{threads}_errs_step{step}"] = np.abs(other[f"errs{step}"] - base[f"errs{step}"])
                out[f"{tag}_t{threads}_cat_step{step}"] = np.array([np.abs(other[f"cat{step}"] - base[f"cat{step}"]).max()])
            out[f"{tag}_t{threads}_eval_pose"] = np.array([np.abs(other["eval_pose"] - base["eval_pose"]).max()])
        out[f"{tag}_pose_scale"] = np.array([np.abs(base["pose3"]).max()])
    torch.set_num_threads(8)
    np.savez_compressed(os.path.join(GOLD, "wrapper_step_spread.npz"), **out)
    for k in sorted(out):
        print(k, out[k])
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


def gen_procrustes():
    import utils.util as U

    s1 = torch.from_numpy(synth_input("procrustes_s1", (6, 16, 3), -30.0, 30.0))
    s2 = torch.from_numpy(synth_input("procrustes_s2", (6, 16, 3), -30.0, 30.0))
    s2[:3] = s1[:3] * 1.7 + 0.3 * s2[:3]          # partly correlated pairs (well-conditioned rotations)
    out = U.batch_compute_similarity_transform_torch(s1, s2)
    np.savez_compressed(os.path.join(GOLD, "procrustes.npz"), s1_hat=out.numpy())
    print("procrustes ok", tuple(out.shape))
    # utils/util.py:337: a BATCH of 2 or 3 frames skips the transpose (the test is meant for unbatched 3 x N / 2 x N points), so the
    # similarity transform is solved over the wrong axes: J "coordinates", 3 "points".  test.py / train_evaluate print exactly these
    # numbers for a ragged last batch of 2 or 3 frames; the drop-in reproduces them by default.
    q = {}
    for B in (2, 3):
        for J in (16, 17):
            a = torch.from_numpy(synth_input(f"procrustes_q1_{B}_{J}", (B, J, 3), -30.0, 30.0))
            b = torch.from_numpy(synth_input(f"procrustes_q2_{B}_{J}", (B, J, 3), -30.0, 30.0))
            b[:1] = a[:1] * 1.3 + 0.5 * b[:1]
            q[f"s1_hat_b{B}_j{J}"] = U.batch_compute_similarity_transform_torch(a, b).numpy()
    np.savez_compressed(os.path.join(GOLD, "procrustes_batch_axes.npz"), **q)
    print("procrustes batch-axes quirk ok", {k: v.shape for k, v in q.items()})


def gen_hm():
    """E1-E9: the reference's HeatMap_UnrealEgo_Shared with OUR ResNet-18 restatement
    standing in for torchvision (parity of the backbone itself is unpinned)."""
    import model.net_architecture as na

    for tag, n_pos, n_rot in (("pos", 15, 0), ("rot", 0, 15)):
        opt = make_opt("UnrealEgo")
        opt.num_heatmap, opt.num_rot_heatmap = n_pos, n_rot
        net = na.HeatMap_UnrealEgo_Shared(opt, "resnet18", input_channel_scale=2)
        load_synth(net, salt=f"hm_{tag}.")
        net.eval()
        left = torch.from_numpy(synth_input("rgb_left", (1, 3, 256, 256), -2.0, 2.0))
        right = torch.from_numpy(synth_input("rgb_right", (1, 3, 256, 256), -2.0, 2.0))
        cap = {}

        def hook(name):
            def f(mod, inp, out):
                cap[name] = out
            return f

        ab = net.after_backbone
        hs = [getattr(ab, n).register_forward_hook(hook(n)) for n in
              ("layer4_1x1", "layer3_1x1", "conv_up3", "conv_up2", "conv_up1")]
        hs.append(net.backbone.backbone.register_forward_hook(hook("pyramid")))
        with torch.no_grad():
            y = net(left, right)
        for h in hs:
            h.remove()
        ref_sd = net.state_dict()
        out = {"state_keys": np.array(list(ref_sd.keys())),
               "state_shapes": np.array(["x".join(str(d) for d in v.shape) for v in ref_sd.values()]),
               "param_keys": np.array([k for k, _ in net.named_parameters()]),
               "out_sample": sample(y, 97), "out_stats": stats(y), "out_shape": np.array(y.shape),
               "out_ch0": y[0, 0].numpy(), "out_last": y[0, -1].numpy()}
        for k in ("layer4_1x1", "layer3_1x1", "conv_up3", "conv_up2", "conv_up1"):
            out[k + "_sample"] = sample(cap[k], 997)
            out[k + "_stats"] = stats(cap[k])
        for i, t in enumerate(cap["pyramid"][1:], start=0):   # layer0..layer4 of the RIGHT eye (last call)
            out[f"pyr{i}_sample"] = sample(t, 997)
            out[f"pyr{i}_stats"] = stats(t)
        np.savez_compressed(os.path.join(GOLD, f"hm_full_{tag}.npz"), **out)
        print(f"hm_full_{tag}:", tuple(y.shape), float(y.abs().mean()))


def gen_sched():
    """learning-rate tables of the reference's own get_scheduler (model/network.py:35-55) on a dummy optimizer"""
    import types
    from model import network as N

    out = {}
    for policy, kw in (("cos_anneal_warmup", dict(niter=1, niter_decay=15, epoch_iter_cnt=7)),
                       ("cos_anneal_warmup", dict(niter=2, niter_decay=3, epoch_iter_cnt=5)),
                       ("cos_anneal", dict(niter=1, niter_decay=4, epoch_iter_cnt=6)),
                       ("lambda", dict(niter=3, niter_decay=5, epoch_iter_cnt=1)),
                       ("step", dict(niter=1, niter_decay=1, epoch_iter_cnt=1)),
                       ("exponent", dict(niter=1, niter_decay=1, epoch_iter_cnt=1))):
        opt = types.SimpleNamespace(lr_policy=policy, epoch_count=1, lr_decay_iters_step=4, **kw)
        prm = torch.nn.Parameter(torch.zeros(1))
        optim = torch.optim.AdamW([prm], lr=1e-3)
        sch = N.get_scheduler(optim, opt)
        steps = (kw["niter"] + kw["niter_decay"]) * kw["epoch_iter_cnt"] + 3
        lrs = []
        for _ in range(steps):
            lrs.append(optim.param_groups[0]["lr"])
            optim.step()
            sch.step()
        tag = f"{policy}_{kw['niter']}_{kw['niter_decay']}_{kw['epoch_iter_cnt']}"
        out[tag] = np.array(lrs, dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, "lr_schedules.npz"), **out)
    print("lr_schedules ok", {k: len(v) for k, v in out.items()})


def gen_hm_train():
    """One optimize_parameters() of the reference's own stage-1 wrapper (model/heatmap_shared_model.py: train-mode forward of
    HeatMap_UnrealEgo_Shared over this repo's ResNet-18 stand-in, MSE losses, torch.optim.Adam) on CPU: losses, a strided
    sample + norm of every gradient, updated BatchNorm running statistics, updated parameters."""
    from model.heatmap_shared_model import HeatmapSharedModel
    from egotap_amd.synthetic import synth_hm_state_dict

    for tag, nh, nr in (("pos", 15, 0), ("rot", 0, 15)):
        opt = make_opt("UnrealEgo")
        opt.num_heatmap, opt.num_rot_heatmap = nh, nr
        opt.model, opt.isTrain, opt.use_amp, opt.gpu_ids = "heatmap_shared", True, False, []
        opt.log_dir, opt.experiment_name, opt.init_type = "/tmp", "gold_hm", "kaiming"
        opt.path_to_trained_heatmap = None
        opt.lr, opt.weight_decay, opt.lr_policy = 1e-3, 0.0, "cos_anneal_warmup"
        opt.niter, opt.niter_decay, opt.epoch_iter_cnt, opt.epoch_count = 1, 3, 4, 1
        opt.lambda_heatmap, opt.lambda_rot_heatmap, opt.distributed = 1.0, 1.0, False
        m = HeatmapSharedModel()
        m.initialize(opt)
        n = m.net_HeatMap
        sd_np = synth_hm_state_dict(nh + 2 * nr, f"hm_{tag}.")
        n.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
        B, C = 2, nh + 2 * nr
        data = {"input_rgb_left": torch.from_numpy(synth_input(f"tr_rgbL_{tag}", (B, 3, 256, 256), -2.0, 2.0)),
                "input_rgb_right": torch.from_numpy(synth_input(f"tr_rgbR_{tag}", (B, 3, 256, 256), -2.0, 2.0)),
                "gt_local_pose": torch.zeros(B, 16, 3), "gt_limb_theta": torch.zeros(B, 15)}
        gt = torch.from_numpy(synth_input(f"tr_gt_{tag}", (B, 2 * C, 64, 64), 0.0, 1.0))
        plen = torch.from_numpy(synth_input(f"tr_plen_{tag}", (B, 2 * C), 2.0, 40.0))
        if tag == "pos":
            data.update(gt_heatmap_left=gt[:, :C], gt_heatmap_right=gt[:, C:])
        else:
            data.update(gt_heatmap_left=torch.zeros(B, 0, 64, 64), gt_heatmap_right=torch.zeros(B, 0, 64, 64),
                        gt_limb_heatmap_left=gt[:, :C], gt_limb_heatmap_right=gt[:, C:], gt_plength_left=plen[:, :C], gt_plength_right=plen[:, C:])
        m.set_input(data)
        m.optimize_parameters()
        out = {"pred_sample": sample(m.pred_heatmap_cat, 97), "pred_stats": stats(m.pred_heatmap_cat)}
        for ln in m.loss_names:
            out["loss_" + ln] = getattr(m, "loss_" + ln).detach().numpy()
        names, norms = [], []
        for k, prm in n.named_parameters():
            if prm.grad is None:
                continue
            names.append(k)
            norms.append(float(prm.grad.double().norm()))
            out["g:" + k] = prm.grad.reshape(-1)[:: max(1, prm.numel() // 257)].numpy().copy()
            out["p:" + k] = prm.detach().reshape(-1)[:: max(1, prm.numel() // 257)].numpy().copy()
        out["grad_keys"], out["grad_norms"] = np.array(names), np.array(norms)
        for k, v in n.state_dict().items():
            if k.startswith("backbone.backbone.backbone.") and (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")):
                out["buf:" + k] = v.numpy().copy()
        out["lr_after"] = np.array([m.optimizers[0].param_groups[0]["lr"]])
        np.savez_compressed(os.path.join(GOLD, f"hm_train_step_{tag}.npz"), **out)
        print(f"hm_train_step_{tag}:", {ln: float(getattr(m, "loss_" + ln)) for ln in m.loss_names}, len(names), "gradients")


def gen_hm_train_b8():
    """A better-conditioned pin of the stage-1 step: the same reference wrapper step as gen_hm_train for the
    position net, but B = 8 (BatchNorm statistics over 8 x 2 eyes) and computed TWICE -- in float64 (the fixture's values) and in the
    reference's own fp32 (recorded only as its deviation from the float64 run: what fp32 arithmetic of this network costs whoever
    does it, i.e. the floor of any fp32 implementation's gate)."""
    from model.heatmap_shared_model import HeatmapSharedModel
    from egotap_amd.synthetic import synth_hm_state_dict

    nh, nr, B = 15, 0, 8
    C = nh

    def run(dtype):
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            opt = make_opt("UnrealEgo")
            opt.num_heatmap, opt.num_rot_heatmap = nh, nr
            opt.model, opt.isTrain, opt.use_amp, opt.gpu_ids = "heatmap_shared", True, False, []
            opt.log_dir, opt.experiment_name, opt.init_type = "/tmp", "gold_hm", "kaiming"
            opt.path_to_trained_heatmap = None
            opt.lr, opt.weight_decay, opt.lr_policy = 1e-3, 0.0, "cos_anneal_warmup"
            opt.niter, opt.niter_decay, opt.epoch_iter_cnt, opt.epoch_count = 1, 3, 4, 1
            opt.lambda_heatmap, opt.lambda_rot_heatmap, opt.distributed = 1.0, 1.0, False
            m = HeatmapSharedModel()
            m.initialize(opt)
            n = m.net_HeatMap
            n.to(dtype)
            sd_np = synth_hm_state_dict(C, "hm_pos.")
            n.load_state_dict({k: torch.from_numpy(v).to(dtype) if v.dtype.kind == "f" else torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
            data = {"input_rgb_left": torch.from_numpy(synth_input("tr8_rgbL", (B, 3, 256, 256), -2.0, 2.0)).to(dtype),
                    "input_rgb_right": torch.from_numpy(synth_input("tr8_rgbR", (B, 3, 256, 256), -2.0, 2.0)).to(dtype),
                    "gt_local_pose": torch.zeros(B, 16, 3, dtype=dtype), "gt_limb_theta": torch.zeros(B, 15, dtype=dtype)}
            gt = torch.from_numpy(synth_input("tr8_gt", (B, 2 * C, 64, 64), 0.0, 1.0)).to(dtype)
            data.update(gt_heatmap_left=gt[:, :C], gt_heatmap_right=gt[:, C:])
            m.set_input(data)
            m.optimize_parameters()
            res = {"pred": m.pred_heatmap_cat.detach().double(), "loss": {ln: float(getattr(m, "loss_" + ln)) for ln in m.loss_names}, "grads": {}}
            for k, prm in n.named_parameters():
                if prm.grad is not None:
                    res["grads"][k] = prm.grad.detach().double().clone()
            res["bufs"] = {k: v.detach().double().clone() for k, v in n.state_dict().items()
                           if k.startswith("backbone.backbone.backbone.") and (k.endswith("running_mean") or k.endswith("running_var"))}
            return res
        finally:
            torch.set_default_dtype(old)

    r64, r32 = run(torch.float64), run(torch.float32)
    out = {"pred_sample": r64["pred"].reshape(-1)[::97].numpy().copy()}
    for ln, v in r64["loss"].items():
        out["loss_" + ln] = np.array(v)
    names, norms, dev_norm, dev_samp = [], [], [], []
    for k, g in r64["grads"].items():
        names.append(k)
        nrm = float(g.norm())
        norms.append(nrm)
        st = max(1, g.numel() // 257)
        out["g:" + k] = g.reshape(-1)[::st].numpy().copy()
        g32 = r32["grads"][k]
        scale = max(nrm / np.sqrt(g.numel()), 1e-300)
        dev_norm.append(abs(float(g32.norm()) - nrm) / max(nrm, 1e-300))
        dev_samp.append(float((g32.reshape(-1)[::st] - g.reshape(-1)[::st]).abs().max()) / scale)
    out["grad_keys"], out["grad_norms"] = np.array(names), np.array(norms)
    out["ref_fp32_dev_norm"], out["ref_fp32_dev_sample"] = np.array(dev_norm), np.array(dev_samp)
    for k, v in r64["bufs"].items():
        out["buf:" + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "hm_train_step_pos_b8.npz"), **out)
    print("hm_train_step_pos_b8:", r64["loss"], len(names), "gradients; reference fp32 vs float64: worst norm deviation",
          f"{max(dev_norm):.3e}, worst sample deviation {max(dev_samp):.3e} of the tensor's typical magnitude")


def gen_synth():
    """ground-truth heatmap synthesis by the reference's own coord2d_to_heatmap / get_limb_data; skimage.draw.line_aa (not
    installed) is supplied by oracle/heatmap_synth_ref.line_aa, so that one step is NOT pinned by this fixture"""
    from oracle import heatmap_synth_ref as R
    sys.modules["skimage.draw"].line_aa = R.line_aa
    import importlib
    import utils.data as UD
    import utils.projection as UP
    importlib.reload(UD)                       # rebinds `from skimage.draw import line_aa`
    out = {}
    for preset, n in (("UnrealEgo", 16), ("EgoCap", 18)):
        for res in (64, 128):
            # joints spread over (and slightly beyond) the 1024-pixel frame, one exactly on a pixel centre, one off-frame
            p2l = synth_input(f"synth_p2l_{preset}", (3, n, 2), -60.0, 1080.0).astype(np.float64)
            p2r = synth_input(f"synth_p2r_{preset}", (3, n, 2), -60.0, 1080.0).astype(np.float64)
            p2l[0, 1] = [512.0, 256.0]
            p2l[0, 2] = [-100.0, 500.0]
            p3 = synth_input(f"synth_p3_{preset}", (3, n, 3), -40.0, 40.0).astype(np.float64)
            for b in range(3):
                hl = UP.coord2d_to_heatmap(p2l[b][1:], res=res, sigma=1.0)
                hr = UP.coord2d_to_heatmap(p2r[b][1:], res=res, sigma=1.0)
                ll, len_l, th = UD.get_limb_data(p2l[b].copy(), p3[b], res=res, area=res, htype="line", sigma=1, joint_preset=preset)
                lr, len_r, _ = UD.get_limb_data(p2r[b].copy(), p3[b], res=res, area=res, htype="line", sigma=1, joint_preset=preset)
                tag = f"{preset}_{res}_{b}"
                out[tag + "_pos"] = np.concatenate([hl, hr]).astype(np.float32)
                out[tag + "_limb"] = np.concatenate([ll, lr]).astype(np.float32)
                out[tag + "_len"] = np.stack([len_l, len_r]).astype(np.float32)
                out[tag + "_theta"] = th.astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "heatmap_synth.npz"), **out)
    print("heatmap_synth ok", len(out), "arrays")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="lift,pu,fc,loss,hm,procrustes,train,sched,synth,hmtrain,wrapper,wrapper_rgb")
    args = ap.parse_args()
    which = set(args.only.split(","))
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    apply_shims()
    if "lift" in which:
        gen_lift("ue", "UnrealEgo")
        gen_lift("ec", "EgoCap")
    if "pu" in which:
        gen_pu()
    if "fc" in which:
        gen_fcblock()
    if "loss" in which:
        gen_loss()
    if "hm" in which:
        gen_hm()
    if "procrustes" in which:
        gen_procrustes()
    if "train" in which:
        gen_train()
    if "sched" in which:
        gen_sched()
    if "synth" in which:
        gen_synth()
    if "hmtrain" in which:
        gen_hm_train()
    if "hmtrain8" in which:
        gen_hm_train_b8()
    if "wrapper" in which:
        gen_wrapper()
    if "wrapper_rgb" in which:
        gen_wrapper_rgb()
    if "wrapper_spread" in which:      # (not in the default set: six reference wrapper runs, ~10 min)
        gen_wrapper_spread()


if __name__ == "__main__":
    main()
