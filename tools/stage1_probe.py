#!/usr/bin/env python3
"""GPU probe: one stage-1 training step of the position heatmap estimator at batch B (for rocprofv3 --stats)."""
import sys, os, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from egotap_amd import models
from egotap_amd.options import preset_defaults
from egotap_amd.synthetic import synth_hm_state_dict, synth_input
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MODE = sys.argv[2] if len(sys.argv) > 2 else "f32"
opt = preset_defaults("UnrealEgo")
opt.model, opt.isTrain, opt.gpu_ids, opt.num_rot_heatmap, opt.lr, opt.weight_decay = "heatmap_shared", True, [0], 0, 1e-3, 0.0
m = models.create_model(opt)
m.net_HeatMap.load_state_dict({k: torch.from_numpy(v) for k, v in synth_hm_state_dict(15, "hm_pos.").items()})
rep = B // 8
data = {"input_rgb_left": torch.from_numpy(synth_input("s1_l", (8, 3, 256, 256), -2.0, 2.0)).cuda().repeat(rep, 1, 1, 1),
        "input_rgb_right": torch.from_numpy(synth_input("s1_r", (8, 3, 256, 256), -2.0, 2.0)).cuda().repeat(rep, 1, 1, 1),
        "gt_heatmap_left": torch.from_numpy(synth_input("s1_gl", (8, 15, 64, 64))).cuda().repeat(rep, 1, 1, 1),
        "gt_heatmap_right": torch.from_numpy(synth_input("s1_gr", (8, 15, 64, 64))).cuda().repeat(rep, 1, 1, 1)}
m.net_HeatMap.set_precision(MODE)
m.set_input(data)
m.optimize_parameters(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2):
    m.optimize_parameters()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 2
print(json.dumps({"B": B, "mode": MODE, "ms_per_step": round(dt * 1e3, 1), "frames_per_s": round(B / dt, 1), "loss": m.get_current_errors()}))
