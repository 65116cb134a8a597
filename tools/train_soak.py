import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_configs import _model, _data
for amp in (False, True):
    m, p = _model(use_amp=amp)
    data, hm, gt = _data(64, p, "soak", gt_range=1.0)
    m.set_input(data)
    losses = []
    for it in range(40):
        m.optimize_parameters()
        if it % 5 == 0 or it == 39:
            e = m.get_current_errors()
            losses.append(round(e["pose"], 4))
    print("use_amp", amp, losses)

# [r5] from RGB with the frozen estimators as train.py runs them (train mode: batch-statistics BatchNorm on the bf16 channels-last kernels, running
# statistics drifting) under --use_amp: 40 steps on one batch -- the loss must fall, everything must stay finite, the estimators' parameters must not move
import os, tempfile
from egotap_amd import models, spec
from egotap_amd.options import preset_defaults
from egotap_amd.synthetic import synth_hm_state_dict, synth_input, synth_state_dict
tmp = tempfile.mkdtemp()
for sub, nh, salt in (("hm_pos", 15, "hm_pos."), ("hm_sin", 30, "hm_rot.")):
    os.makedirs(os.path.join(tmp, sub))
    torch.save({k: torch.from_numpy(v) for k, v in synth_hm_state_dict(nh, salt).items()}, os.path.join(tmp, sub, "best_net_HeatMap.pth"))
opt = preset_defaults("UnrealEgo")
opt.gpu_ids, opt.isTrain, opt.use_gt_heatmap, opt.use_amp = [0], True, False, True
opt.lr, opt.opt_eps, opt.weight_decay, opt.log_dir = 1e-3, 1e-4, 0.0, tmp
opt.path_to_trained_heatmap = os.path.join(tmp, "hm", "best_net_HeatMap.pth")
m = models.create_model(opt)
p = spec.lift_preset("UnrealEgo")
m.net_AutoEncoder.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(p)).items()})
m.train()
B = 16
m.set_input({"input_rgb_left": torch.from_numpy(synth_input("soak_l", (B, 3, 256, 256), -2.0, 2.0)), "input_rgb_right": torch.from_numpy(synth_input("soak_r", (B, 3, 256, 256), -2.0, 2.0)),
             "gt_local_pose": torch.from_numpy(synth_input("soak_gt", (B, 16, 3), -1.0, 1.0))})
w0 = {k: v.clone() for k, v in m.net_HeatMap.named_parameters()}
losses = []
for it in range(40):
    m.optimize_parameters()
    if it % 5 == 0 or it == 39:
        losses.append(round(m.get_current_errors()["pose"], 4))
torch.cuda.synchronize()
ok = all(torch.isfinite(v).all() for v in m.net_HeatMap.state_dict().values()) and all(torch.equal(v, w0[k]) for k, v in m.net_HeatMap.named_parameters())
print("from RGB, reference-default estimators, use_amp:", losses, "buffers finite and parameters frozen:", bool(ok),
      "batches tracked:", int(m.net_HeatMap.state_dict()["backbone.backbone.backbone.bn1.num_batches_tracked"]))
assert ok and losses[-1] < losses[0] and all(x == x for x in losses)
