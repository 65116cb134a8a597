#!/usr/bin/env python3
"""GPU probe: full pipeline (two estimators + lifting head) from RGB at B per GPU, fp32 vs bf16x3: step time and pose difference."""
import sys, os, json, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from egotap_amd import models, spec
from egotap_amd.options import preset_defaults
from egotap_amd.synthetic import synth_hm_state_dict, synth_input, synth_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
opt = preset_defaults("UnrealEgo"); opt.gpu_ids = [0]; opt.isTrain = False
m = models.create_model(opt)
p = spec.lift_preset("UnrealEgo")
m.net_AutoEncoder.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(p)).items()})
m.net_HeatMap.load_state_dict({k: torch.from_numpy(v) for k, v in synth_hm_state_dict(m.net_HeatMap.num_heatmap, "hm_pos.").items()})
m.net_RotHeatMap.load_state_dict({k: torch.from_numpy(v) for k, v in synth_hm_state_dict(m.net_RotHeatMap.num_heatmap, "hm_rot.").items()})
m.set_eval_mode()
rl = torch.from_numpy(synth_input("rgb_l", (8, 3, 256, 256), -2.0, 2.0)).cuda().repeat(B // 8, 1, 1, 1).contiguous()
rr = torch.from_numpy(synth_input("rgb_r", (8, 3, 256, 256), -2.0, 2.0)).cuda().repeat(B // 8, 1, 1, 1).contiguous()
m.set_input({"input_rgb_left": rl, "input_rgb_right": rr})
out = {}
for mode in ("f32", "bf16x3"):
    m.set_precision(mode)
    with torch.no_grad():
        m.forward(evaluate=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            m.forward(evaluate=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    out[mode] = (m.pred_pose.clone(), m.pred_heatmap_cat[:8].clone())
    print(json.dumps({"B": B, "mode": mode, "ms_per_step": round(dt * 1e3, 2), "frames_per_s": round(B / dt, 1)}), flush=True)
print(json.dumps({"max_abs_pose_diff": float((out["f32"][0] - out["bf16x3"][0]).abs().max()),
                  "max_abs_heatmap_diff": float((out["f32"][1] - out["bf16x3"][1]).abs().max()),
                  "heatmap_abs_max": float(out["f32"][1].abs().max())}))
