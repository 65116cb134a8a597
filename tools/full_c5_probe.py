#!/usr/bin/env python3
"""GPU probe: BASELINE config 5 on one GPU -- 512 x 512 stereo RGB -> two EgoCap estimators (128 x 128 maps) -> head, through the wrapper's
evaluation forward, batch B, arithmetic `mode` (for rocprofv3 --stats).  usage: full_c5_probe.py B mode"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from egotap_amd import models, spec
from egotap_amd.options import preset_defaults
from egotap_amd.synthetic import synth_hm_state_dict, synth_input, synth_state_dict
B, mode = int(sys.argv[1]), sys.argv[2]
p5 = spec.lift_preset("EgoCap", 128)
opt = preset_defaults("EgoCap", 128)
opt.gpu_ids = [0]
m = models.create_model(opt)
J = p5.n_joints_hm
for name, sd in (("AutoEncoder", synth_state_dict(spec.lift_state_spec(p5))), ("HeatMap", synth_hm_state_dict(J, "hm_pos.")),
                 ("RotHeatMap", synth_hm_state_dict(2 * J, "hm_rot."))):
    getattr(m, "net_" + name).load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
rgb = [torch.from_numpy(synth_input(f"rgb5p_{s}", (4, 3, 512, 512), -2.0, 2.0)).cuda().repeat(B // 4, 1, 1, 1).contiguous() for s in ("l", "r")]
m.set_input({"input_rgb_left": rgb[0], "input_rgb_right": rgb[1]})
m.set_eval_mode()
m.set_precision(mode)
with torch.no_grad():
    for _ in range(2):
        m.forward(evaluate=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m.forward(evaluate=True)
    torch.cuda.synchronize()
print(f"config-5 full pipeline B={B} {mode}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per forward = {B * 5 / (time.perf_counter() - t0):.1f} frames/s")
