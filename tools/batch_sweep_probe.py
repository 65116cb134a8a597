#!/usr/bin/env python3
"""GPU probe: the lifting head at the batch sizes where the GEMM routing changes (split-K thresholds), against the B = 2 result."""
import sys, os, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from egotap_amd.synthetic import synth_input
from gpu_util import lift_net
for preset, tag in (("UnrealEgo", "ue"), ("EgoCap", "ec")):
    net, _, p = lift_net(preset)
    g = np.load(os.path.join(REPO, "tests", "golden", f"lift_fwd_{tag}_b2.npz"))
    two = torch.from_numpy(synth_input(f"hm_{tag}", (2, p.in_channels, 64, 64))).cuda()
    for mode in ("f32", "bf16x3"):
        net.set_precision(mode)
        worst = 0.0
        for B in (1, 2, 3, 7, 29, 30, 31, 34, 35, 36, 63, 106, 107, 108, 129):
            x = two.repeat((B + 1) // 2, 1, 1, 1)[:B].contiguous()
            out = net.predict_pose(x).cpu().numpy()
            ref = np.tile(g["pose"], ((B + 1) // 2, 1, 1))[:B]
            worst = max(worst, float(np.abs(out - ref).max()))
        print(json.dumps({"preset": preset, "mode": mode, "max_abs_vs_reference_golden_over_batch_sizes": worst}), flush=True)
        assert worst < 1e-4
