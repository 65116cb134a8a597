"""GPU parity of the whole hot path through the wrapper mirror (RGB -> two heatmap estimators -> lifting head ->
joints + metrics) against the oracle chain on the same synthetic inputs."""
import numpy as np
import pytest
import torch

from egotap_amd.synthetic import synth_hm_state_dict, synth_input, synth_state_dict

pytestmark = pytest.mark.gpu


class _Avg(dict):
    def update(self, d):
        for k, v in d.items():
            self.setdefault(k, []).append(float(v))


def _model():
    from egotap_amd import models, spec
    from egotap_amd.options import preset_defaults
    opt = preset_defaults("UnrealEgo")
    m = models.create_model(opt)
    p = spec.lift_preset("UnrealEgo")
    sds = {"AutoEncoder": synth_state_dict(spec.lift_state_spec(p)),
           "HeatMap": synth_hm_state_dict(15, "hm_pos."), "RotHeatMap": synth_hm_state_dict(30, "hm_rot.")}
    for name, sd in sds.items():
        getattr(m, "net_" + name).load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m, sds, p


def test_wrapper_evaluate_matches_oracle_chain():
    from oracle import hm_ref as H, lift_ref as O
    m, sds, p = _model()
    B = 2
    data = {"input_rgb_left": torch.from_numpy(synth_input("w_rgb_l", (B, 3, 256, 256), -2.0, 2.0)),
            "input_rgb_right": torch.from_numpy(synth_input("w_rgb_r", (B, 3, 256, 256), -2.0, 2.0)),
            "gt_local_pose": torch.from_numpy(synth_input("w_gt", (B, 16, 3), -1.0, 1.0))}
    m.set_input(data)
    m.eval()                                 # test.py -> utils/evaluate.py:93 (set_eval_mode() alone leaves net_RotHeatMap in its mode, as the reference's)
    avg = _Avg()
    pose, hm_cat, avg = m.evaluate(avg)
    torch.cuda.synchronize()
    with torch.no_grad():
        pos = H.hm_forward(data["input_rgb_left"], data["input_rgb_right"], H.to_torch_sd(sds["HeatMap"]))
        rot = H.hm_forward(data["input_rgb_left"], data["input_rgb_right"], H.to_torch_sd(sds["RotHeatMap"]))
        cat = torch.cat([pos, rot], dim=1)
        ref = O.lift_forward(cat, O.to_torch_sd(sds["AutoEncoder"]), p)
        # a batch of TWO frames: the reference's Procrustes step takes its no-transpose branch (utils/util.py:337), reproduced by default
        aligned = O.procrustes_align_batch_axes(ref.double(), data["gt_local_pose"].double()).float()
    assert tuple(hm_cat.shape) == (B, 90, 64, 64)
    np.testing.assert_allclose(hm_cat.cpu().numpy(), cat.numpy(), atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(pose.cpu().numpy(), ref.numpy(), atol=1e-4, rtol=0)
    ref_mpjpe = [float(torch.linalg.norm(data["gt_local_pose"][i] - ref[i], dim=-1).mean() * 10) for i in range(B)]
    ref_pa = [float(torch.linalg.norm(data["gt_local_pose"][i] - aligned[i], dim=-1).mean() * 10) for i in range(B)]
    np.testing.assert_allclose(avg["mpjpe"], ref_mpjpe, rtol=1e-4)
    np.testing.assert_allclose(avg["pa_mpjpe"], ref_pa, rtol=1e-3)
    # attributes the reference's callers read
    assert m.pred_heatmap_left.shape == (B, 15, 64, 64) and m.pred_limb_heatmap_right.shape == (B, 30, 64, 64)
    assert float(m.pred_heatmap_rec_cat.abs().max()) == 0.0 and m.eval_key == "mpjpe"
    with pytest.raises(RuntimeError):
        m.optimize_parameters()              # created with isTrain = False


def test_wrapper_gt_heatmap_path():
    m, sds, p = _model()
    m.opt.use_gt_heatmap = True
    B = 3
    hm = torch.from_numpy(synth_input("w_hm", (B, 90, 64, 64)))
    data = {"input_rgb_left": torch.zeros(B, 3, 256, 256), "input_rgb_right": torch.zeros(B, 3, 256, 256),
            "gt_heatmap_left": hm[:, :15], "gt_heatmap_right": hm[:, 15:30], "gt_limb_heatmap_left": hm[:, 30:60],
            "gt_limb_heatmap_right": hm[:, 60:], "gt_local_pose": torch.zeros(B, 16, 3)}
    m.set_input(data)
    m.set_eval_mode()
    with torch.no_grad():
        m.forward(evaluate=True)
    direct = m.net_AutoEncoder.predict_pose(hm.cuda())
    torch.cuda.synchronize()
    assert torch.equal(m.pred_pose, direct)
    cat_copy = m.pred_heatmap_cat
    # [r5] maps that already ARE consecutive channel slices of one resident tensor in the head's layout: the concatenation is a view of it (no copy),
    # same values and same pose; any other arrangement (a different order, a gap, another dtype) still goes through torch.cat
    base = hm.cuda()
    views = {"gt_heatmap_left": base[:, :15], "gt_heatmap_right": base[:, 15:30], "gt_limb_heatmap_left": base[:, 30:60], "gt_limb_heatmap_right": base[:, 60:]}
    m.set_input(dict(data, **views))
    with torch.no_grad():
        m.forward(evaluate=True)
    torch.cuda.synchronize()
    assert m.pred_heatmap_cat.data_ptr() == base.data_ptr() and torch.equal(m.pred_heatmap_cat, cat_copy) and torch.equal(m.pred_pose, direct)
    m.set_input(dict(data, **dict(views, gt_heatmap_left=base[:, 15:30], gt_heatmap_right=base[:, :15])))      # swapped eyes: not the tensor's own order
    with torch.no_grad():
        m.forward(evaluate=True)
    assert m.pred_heatmap_cat.data_ptr() != base.data_ptr() and torch.equal(m.pred_heatmap_cat[:, :15], base[:, 15:30])


def test_wrapper_synthesises_gt_heatmaps_from_joints():
    """--use_gt_heatmap with joints in the batch instead of rendered heatmaps: the wrapper renders them on the device, in the
    head's layout, and the pose equals the one from CPU-rendered heatmaps (oracle/heatmap_synth_ref.py) fed the usual way."""
    import numpy as np
    from egotap_amd import models, spec
    from egotap_amd.options import preset_defaults
    from egotap_amd.synthetic import synth_input, synth_state_dict
    from oracle import heatmap_synth_ref as R
    opt = preset_defaults("UnrealEgo")
    opt.gpu_ids, opt.use_gt_heatmap = [0], True
    m = models.create_model(opt)
    p = spec.lift_preset("UnrealEgo")
    m.net_AutoEncoder.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(p)).items()})
    m.set_eval_mode()
    B = 2
    p2l, p2r = synth_input("wr_p2l", (B, 16, 2), 0.0, 1024.0), synth_input("wr_p2r", (B, 16, 2), 0.0, 1024.0)
    p3 = synth_input("wr_p3", (B, 16, 3), -40.0, 40.0)
    rgb = torch.zeros(B, 3, 256, 256)
    m.set_input({"input_rgb_left": rgb, "input_rgb_right": rgb, "gt_camera_2d_left": torch.from_numpy(p2l),
                 "gt_camera_2d_right": torch.from_numpy(p2r), "gt_local_pose": torch.from_numpy(p3)})
    with torch.no_grad():
        m.forward(evaluate=True)
    pose_dev = m.pred_pose.clone()
    cats = np.stack([R.process_frame(p2l[b].astype(np.float64), p2r[b].astype(np.float64), p3[b].astype(np.float64))[0] for b in range(B)])
    cat = torch.from_numpy(cats)
    m.set_input({"input_rgb_left": rgb, "input_rgb_right": rgb, "gt_heatmap_left": cat[:, :15], "gt_heatmap_right": cat[:, 15:30],
                 "gt_limb_heatmap_left": cat[:, 30:60], "gt_limb_heatmap_right": cat[:, 60:], "gt_local_pose": torch.from_numpy(p3)})
    m._gt_cat = None
    with torch.no_grad():
        m.forward(evaluate=True)
    assert float((m.pred_pose - pose_dev).abs().max()) < 1e-4
