#!/usr/bin/env python3
"""rehearsal: stage-1 wrapper steps under torch.distributed (gloo or nccl), prints the loss after every step on each rank"""
import sys, os, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from egotap_amd import models, parallel
from egotap_amd.options import preset_defaults
from egotap_amd.synthetic import synth_hm_state_dict, synth_input
rank = int(os.environ.get("RANK", "0"))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
parallel.init_from_env(os.environ.get("EGOTAP_DIST_BACKEND", "gloo"), dev)
B = 8
opt = preset_defaults("UnrealEgo")
opt.model, opt.isTrain, opt.gpu_ids, opt.num_rot_heatmap, opt.lr, opt.weight_decay = "heatmap_shared", True, [0], 0, 1e-3, 0.0
m = models.create_model(opt)
m.net_HeatMap.load_state_dict({k: torch.from_numpy(v) for k, v in synth_hm_state_dict(15, "hm_pos.").items()})
data = {"input_rgb_left": torch.from_numpy(synth_input(f"s1_l_rank{rank}", (B, 3, 256, 256), -2.0, 2.0)).cuda(),
        "input_rgb_right": torch.from_numpy(synth_input(f"s1_r_rank{rank}", (B, 3, 256, 256), -2.0, 2.0)).cuda(),
        "gt_heatmap_left": torch.from_numpy(synth_input(f"s1_gl_rank{rank}", (B, 15, 64, 64))).cuda(),
        "gt_heatmap_right": torch.from_numpy(synth_input(f"s1_gr_rank{rank}", (B, 15, 64, 64))).cuda()}
m.set_input(data)
for step in range(3):
    m.optimize_parameters()
    torch.cuda.synchronize()
    g = sum(float(p.grad.double().norm()) ** 2 for p in m.net_HeatMap.parameters() if p.grad is not None) ** 0.5
    print(json.dumps({"rank": rank, "step": step, "loss": m.get_current_errors(), "grad_norm": g}), flush=True)
