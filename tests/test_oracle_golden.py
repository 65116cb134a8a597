"""Pins the oracle (oracle/lift_ref.py) against golden vectors captured from the
reference's own modules (tools/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from egotap_amd import spec
from egotap_amd.synthetic import synth_state_dict, synth_input, synth_tensor
from oracle import lift_ref as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def _sample(t, stride=997):
    return t.reshape(-1)[::stride].numpy()


@pytest.mark.parametrize("tag,preset", [("ue", "UnrealEgo"), ("ec", "EgoCap")])
def test_state_spec_matches_reference(tag, preset):
    g = _load(f"lift_fwd_{tag}_b2.npz")
    p = spec.lift_preset(preset)
    ours = spec.lift_state_spec(p)
    assert [k for k, _ in ours] == list(g["state_keys"])
    assert ["x".join(str(d) for d in shp) for _, shp in ours] == list(g["state_shapes"])
    params = [k for k, _ in ours if not spec.is_buffer(k)]
    assert params == list(g["param_keys"])
    n_params = sum(int(np.prod(shp)) for k, shp in ours if not spec.is_buffer(k))
    assert n_params == {"ue": 96_984_585, "ec": 96_938_499}[tag] or n_params > 96_000_000


@pytest.mark.parametrize("tag,preset", [("ue", "UnrealEgo"), ("ec", "EgoCap")])
def test_lift_forward_matches_reference(tag, preset):
    g = _load(f"lift_fwd_{tag}_b2.npz")
    p = spec.lift_preset(preset)
    sd = O.to_torch_sd(synth_state_dict(spec.lift_state_spec(p)))
    hm = torch.from_numpy(synth_input(f"hm_{tag}", (2, p.in_channels, 64, 64)))
    trace = {}
    torch.set_num_threads(8)
    with torch.no_grad():
        pose = O.lift_forward(hm, sd, p, trace)
    assert tuple(pose.shape) == (2, p.out_joints, 3)
    np.testing.assert_allclose(pose.numpy(), g["pose"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(trace["pos_embed"].numpy(), g["pos_embed"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(trace["rot_embed"].numpy(), g["rot_embed"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(trace["skel_embed"].numpy(), g["skel_embed"], atol=1e-5, rtol=0)
    for k in ("emb", "layer0", "layer1", "layer2", "final_ln"):
        np.testing.assert_allclose(_sample(trace[k]), g[k + "_sample"], atol=2e-5, rtol=1e-5)
        s = trace[k].double()
        np.testing.assert_allclose([s.sum().item(), s.abs().sum().item()], g[k + "_stats"], rtol=1e-5)
    assert bool(g["rot_is_zero"]) and bool(g["indep_is_zero"]) and bool(g["out_hm_is_zero"])


@pytest.mark.parametrize("tag,preset", [("ue", "UnrealEgo"), ("ec", "EgoCap")])
def test_pu_chain_is_chain_not_tree(tag, preset):
    g = _load(f"pu_chain_{tag}.npz")
    p = spec.lift_preset(preset)
    full = spec.lift_state_spec(p)
    sd = O.to_torch_sd(synth_state_dict([(k, s) for k, s in full if k.startswith("skel_sequential_layer.")]))
    J = p.n_joints_hm
    x = torch.from_numpy(synth_input(f"pu_x_{tag}", (J, 3, 256), -1.0, 1.0))
    b = torch.from_numpy(synth_input(f"pu_b_{tag}", (J, 3, 256), -1.0, 1.0))
    chain = O.pu_chain(x, b, sd)
    np.testing.assert_allclose(chain.numpy(), g["out"], atol=1e-6, rtol=0)
    parents = O.UE_PARENTS if preset == "UnrealEgo" else O.EC_PARENTS
    tree = O.pu_chain(x, b, sd, tree_parents=parents)
    assert tree.shape == chain.shape
    assert float((tree - torch.from_numpy(g["out"])).abs().max()) > 1e-2   # the tree is the WRONG answer


def test_fc_block_eval_and_train():
    g = _load("fcblock.npz")
    keys = [("fcblock.fc.weight", (64, 96)), ("fcblock.fc.bias", (64,)), ("fcblock.bn.weight", (64,)),
            ("fcblock.bn.bias", (64,)), ("fcblock.bn.running_mean", (64,)), ("fcblock.bn.running_var", (64,))]
    sd = {k: torch.from_numpy(synth_tensor(k, s)) for k, s in keys}
    x = torch.from_numpy(synth_input("fcblock_x", (24, 96), -1.0, 1.0))
    y = O.fc_block(x, sd, "fcblock")
    np.testing.assert_allclose(y.numpy(), g["y_eval"], atol=1e-6)
    y, (rm, rv) = O.fc_block(x, sd, "fcblock", training=True)
    np.testing.assert_allclose(y.numpy(), g["y_train"], atol=2e-6)
    np.testing.assert_allclose(rm.numpy(), g["running_mean"], atol=1e-6)
    np.testing.assert_allclose(rv.numpy(), g["running_var"], atol=1e-6)
    assert int(g["num_batches_tracked"]) == 1


@pytest.mark.parametrize("tag,preset,nj", [("ue", "UnrealEgo", 16), ("ec", "EgoCap", 17)])
def test_losses(tag, preset, nj):
    g = _load(f"loss_{tag}.npz")
    p = spec.lift_preset(preset)
    pred = torch.from_numpy(synth_input(f"loss_pred_{tag}", (5, nj, 3), -20.0, 20.0)).requires_grad_(True)
    gt = torch.from_numpy(synth_input(f"loss_gt_{tag}", (5, nj, 3), -20.0, 20.0))
    mp, cs = O.loss_mpjpe(pred, gt), O.loss_cos_sim(pred, gt, p)
    np.testing.assert_allclose(mp.item(), g["mpjpe"], rtol=1e-6)
    np.testing.assert_allclose(cs.item(), g["cos_sim"], rtol=1e-5)
    (grad,) = torch.autograd.grad(0.1 * mp + (-0.01) * 0.1 * cs, pred)
    np.testing.assert_allclose(grad.numpy(), g["dtotal_dpred"], atol=1e-7)


def test_procrustes_matches_reference():
    g = _load("procrustes.npz")
    s1 = torch.from_numpy(synth_input("procrustes_s1", (6, 16, 3), -30.0, 30.0))
    s2 = torch.from_numpy(synth_input("procrustes_s2", (6, 16, 3), -30.0, 30.0))
    s2[:3] = s1[:3] * 1.7 + 0.3 * s2[:3]
    np.testing.assert_allclose(O.procrustes_align(s1, s2).numpy(), g["s1_hat"], atol=2e-4, rtol=1e-4)


def test_train_step_matches_reference():
    """oracle training step (train-mode BN, loss, autograd, AdamW) against one optimize step of the reference modules"""
    g = _load("train_step_ue_b2.npz")
    p = spec.lift_preset("UnrealEgo")
    sd = O.to_torch_sd(synth_state_dict(spec.lift_state_spec(p)))
    hm = torch.from_numpy(synth_input("hm_train", (2, 90, 64, 64)))
    gt = torch.from_numpy(synth_input("gt_train", (2, 16, 3), -1.0, 1.0))
    torch.set_num_threads(8)
    out = O.train_step(hm, gt, sd, p)
    np.testing.assert_allclose(out["pose"].numpy(), g["pose"], atol=2e-5)
    np.testing.assert_allclose(out["loss_pose"].item(), g["loss_pose"], rtol=1e-5)
    np.testing.assert_allclose(out["loss_cos_sim"].item(), g["loss_cos_sim"], rtol=1e-3, atol=1e-7)
    assert sorted(k for k, v in out["grads"].items() if v is not None) == sorted(g["grad_keys"])
    assert set(g["no_grad_keys"]) == {"pos_heatmap_encoder.vit.embeddings.cls_token", "pos_heatmap_encoder.vit.pooler.dense.weight",
                                      "pos_heatmap_encoder.vit.pooler.dense.bias"}
    norms = dict(zip(g["grad_keys"], g["grad_norms"]))
    for k, gr in out["grads"].items():
        if gr is None:
            continue
        ref_s = g["g:" + k]
        got_s = gr.reshape(-1)[:: max(1, gr.numel() // 257)].numpy()
        scale = max(norms[k] / np.sqrt(gr.numel()), 1e-12)
        assert np.abs(got_s - ref_s).max() <= 2e-3 * scale + 5e-9, k   # some gradients are mathematically zero (noise ~1e-9)
        # key.bias gradients are mathematically zero (softmax is shift invariant): pure rounding noise, absolute check
        np.testing.assert_allclose(float(gr.double().norm()), norms[k], rtol=2e-4, atol=1e-8, err_msg=k)
        new_s = out["new_params"][k].reshape(-1)[:: max(1, gr.numel() // 257)].numpy()
        np.testing.assert_allclose(new_s, g["p:" + k], atol=2e-5, err_msg=k)
    for pre, (rm, rv) in out["bn"].items():
        np.testing.assert_allclose(rm.detach().numpy(), g["buf:" + pre + ".bn.running_mean"], atol=1e-6)
        np.testing.assert_allclose(rv.detach().numpy(), g["buf:" + pre + ".bn.running_var"], atol=1e-6)


def test_lr_schedules_match_reference():
    """get_scheduler (host logic of the training wrapper) against tables produced by the reference's own get_scheduler
    (model/network.py:35-55; cos_anneal_warmup = transformers' cosine schedule with warm-up) on a dummy optimizer."""
    import types
    from egotap_amd.training import get_scheduler
    g = _load("lr_schedules.npz")
    for tag in g.files:
        policy, niter, niter_decay, per = tag.rsplit("_", 3)
        opt = types.SimpleNamespace(lr_policy=policy, epoch_count=1, lr_decay_iters_step=4, niter=int(niter),
                                    niter_decay=int(niter_decay), epoch_iter_cnt=int(per))
        optim = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
        sch = get_scheduler(optim, opt)
        lrs = []
        for _ in range(len(g[tag])):
            lrs.append(optim.param_groups[0]["lr"])
            optim.step()
            sch.step()
        np.testing.assert_allclose(lrs, g[tag], rtol=1e-12, atol=1e-18, err_msg=tag)
    with pytest.raises(NotImplementedError):
        get_scheduler(optim, types.SimpleNamespace(lr_policy="plateau"))


def test_wrapper_eval_fixture_metrics_and_the_batch_of_two_accident():
    """G7 (the reference wrapper's evaluate()): the oracle's MPJPE / Procrustes restatement reproduces the reference's per-sample
    metrics at B = 4.  For a batch of 2 or 3 frames the reference's batch_compute_similarity_transform_torch skips its transpose
    (utils/util.py:337 tests S1.shape[0] against 3 and 2, meant for unbatched 3 x N input) and aligns the wrong axes: recorded in
    the fixture and REPRODUCED (default of the drop-in since round 4: test.py prints these numbers for a ragged last batch) by
    oracle.procrustes_align_batch_axes, the restatement of that branch."""
    import os
    from egotap_amd.synthetic import synth_input
    from oracle import lift_ref as O
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "wrapper_eval_ue_b4.npz"))
    pose = torch.from_numpy(g["gt_pred_pose"])
    gt = torch.from_numpy(synth_input("wrap_gt_eval", (4, 16, 3), -20.0, 20.0))
    mpjpe = [float(torch.linalg.norm(gt[i] - pose[i], dim=-1).mean() * 10) for i in range(4)]
    al = O.procrustes_align(pose, gt)
    pa = [float(torch.linalg.norm(gt[i] - al[i], dim=-1).mean() * 10) for i in range(4)]
    np.testing.assert_allclose(mpjpe, g["gt_mpjpe"], rtol=1e-5)
    np.testing.assert_allclose(pa, g["gt_pa_mpjpe"], rtol=1e-4)
    np.testing.assert_allclose(g["quirk_b2_gt_mpjpe"], g["gt_mpjpe"][:2], rtol=1e-6)          # MPJPE does not depend on the batch
    assert np.all(np.abs(g["quirk_b2_gt_pa_mpjpe"] - g["gt_pa_mpjpe"][:2]) > 50.0)             # the accident: ~100 mm off
    al2 = O.procrustes_align(pose[:2], gt[:2])
    pa2 = [float(torch.linalg.norm(gt[i] - al2[i], dim=-1).mean() * 10) for i in range(2)]
    np.testing.assert_allclose(pa2, g["gt_pa_mpjpe"][:2], rtol=1e-4)                           # the 3 x 3 solve is batch independent ...
    alq = O.procrustes_align_batch_axes(pose[:2].double(), gt[:2].double())
    paq = [float(torch.linalg.norm(gt[i].double() - alq[i], dim=-1).mean() * 10) for i in range(2)]
    np.testing.assert_allclose(paq, g["quirk_b2_gt_pa_mpjpe"], rtol=1e-4)                      # ... and this is what the reference printed


def test_procrustes_batch_axes_restatement_matches_the_reference():
    """utils/util.py:328-379 called with batches of 2 and 3 frames (the no-transpose branch of line 337), 16 and 17 joints:
    tests/golden/procrustes_batch_axes.npz holds the reference's own outputs"""
    from egotap_amd.synthetic import synth_input
    from oracle import lift_ref as O
    g = _load("procrustes_batch_axes.npz")
    for B in (2, 3):
        for J in (16, 17):
            a = torch.from_numpy(synth_input(f"procrustes_q1_{B}_{J}", (B, J, 3), -30.0, 30.0))
            b = torch.from_numpy(synth_input(f"procrustes_q2_{B}_{J}", (B, J, 3), -30.0, 30.0))
            b[:1] = a[:1] * 1.3 + 0.5 * b[:1]
            got = O.procrustes_align_batch_axes(a.double(), b.double()).numpy()
            np.testing.assert_allclose(got, g[f"s1_hat_b{B}_j{J}"], atol=2e-4, rtol=1e-4)
            # and it is NOT the 3 x 3 alignment
            assert np.abs(got - O.procrustes_align(a.double(), b.double()).numpy()).max() > 1.0


def test_bf16_storage_hook_changes_only_what_it_should():
    """round=None is the pinned restatement (every golden test above runs it); round=Bf16Storage moves the pose by bf16-sized
    amounts and leaves the fp32-only parts (propagation units, pose head given equal inputs) alone"""
    from egotap_amd import spec
    from egotap_amd.synthetic import synth_input, synth_state_dict
    from oracle import lift_ref as O
    p = spec.lift_preset("UnrealEgo")
    sd = O.to_torch_sd(synth_state_dict(spec.lift_state_spec(p)), torch.float64)
    hm = torch.from_numpy(synth_input("hm_ue", (1, p.in_channels, 64, 64))).double()
    with torch.no_grad():
        exact = O.lift_forward(hm, sd, p)
        emu = O.lift_forward(hm, sd, p, round=O.Bf16Storage)
    d = float((exact - emu).abs().max())
    assert 1e-5 < d < 5e-2 * float(exact.abs().max())
    x = torch.randn(7, 5, dtype=torch.float64, requires_grad=True)
    y = O.Bf16Storage.bwd(x * 3.0)
    assert torch.equal(y, x * 3.0)                                      # value untouched ...
    (gx,) = torch.autograd.grad(y.sum() * 1.2345678, x)
    assert torch.equal(gx, O.bf16_round(torch.full_like(x, 1.2345678)) * 3.0)      # ... gradient rounded
    w = O.Bf16Storage.w(x)
    assert torch.equal(w, O.bf16_round(x))
    (gw,) = torch.autograd.grad((w * 1.2345678).sum(), x)
    assert torch.equal(gw, torch.full_like(x, 1.2345678))               # straight through to the fp32 master


def test_oracle_gradients_match_the_reference_wrapper_egocap_step():
    """wrapper_ec.npz (tools/make_golden.py gen_wrapper: one optimize_parameters() of the reference's EgoTAPAutoEncoderModel, EgoCap
    preset, 3 frames = 102 encoder rows, fp32 on CPU) against the float64 oracle: losses, pose and a strided sample of every gradient.
    This is the link tests/test_gpu_wrapper_golden.py::test_egocap_wrapper_step... leans on when it compares the HIP gradients with
    the oracle evaluated on the HIP forward's LeakyReLU branches."""
    g = _load("wrapper_ec.npz")
    p = spec.lift_preset("EgoCap")
    sd = O.to_torch_sd(synth_state_dict(spec.lift_state_spec(p)), torch.float64)
    hm = torch.from_numpy(synth_input("wrap_hm_ec_step", (3, p.in_channels, 64, 64))).double()
    gt = torch.from_numpy(synth_input("wrap_gt_ec_step", (3, p.out_joints, 3), -20.0, 20.0)).double()
    hook = {"masks": {}, "pre": {}}
    ref = O.train_step(hm, gt, sd, p, lrelu=hook)
    np.testing.assert_allclose(ref["pose"].numpy(), g["pred_pose_step1"], atol=2e-5)
    assert sorted(k for k, v in ref["grads"].items() if v is not None) == sorted(g["grad_keys"])
    norms = dict(zip(g["grad_keys"], g["grad_norms"]))
    for k in g["grad_keys"]:
        gr = ref["grads"][k].reshape(-1)
        scale = norms[k] / np.sqrt(gr.numel())
        if scale < 1e-8:          # biases in front of a train-mode BatchNorm: exactly zero in exact arithmetic, rounding noise in fp32
            continue
        err = np.abs(gr[:: max(1, gr.numel() // 257)].numpy() - g["g:" + k]).max()
        assert err <= 1e-3 * scale, f"{k}: {err:.3e} vs typical magnitude {scale:.3e}"
    # the hook with no masks is the pinned function; it reports the LeakyReLU inputs of the six blocks
    assert sorted(hook["pre"]) == sorted(f"{e}_heatmap_encoder.fc{j}" for e in ("pos", "rot") for j in (1, 2, 3))
    assert hook["pre"]["pos_heatmap_encoder.fc1"].shape == (102, 2048)
