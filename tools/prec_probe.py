#!/usr/bin/env python3
"""GPU probe: the lifting head in bf16x3 mode against exact fp32 and the reference golden: error and step time."""
import sys, os, json, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from egotap_amd.synthetic import synth_input
from gpu_util import lift_net

for tag, preset in (("ue", "UnrealEgo"), ("ec", "EgoCap")):
    net, _, p = lift_net(preset)
    g = np.load(os.path.join(REPO, "tests", "golden", f"lift_fwd_{tag}_b2.npz"))
    hm = torch.from_numpy(synth_input(f"hm_{tag}", (2, p.in_channels, 64, 64))).cuda()
    out = {}
    for mode in ("f32", "bf16x3"):
        net.set_precision(mode)
        pose = net.predict_pose(hm); torch.cuda.synchronize()
        out[mode] = pose.cpu().numpy()
        print(json.dumps({"preset": preset, "mode": mode, "max_abs_vs_golden": float(np.abs(out[mode] - g["pose"]).max()),
                          "pose_abs_mean": float(np.abs(g["pose"]).mean())}), flush=True)
    net.set_precision("f32")
net, _, p = lift_net("UnrealEgo")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hm = torch.from_numpy(synth_input("hm_bench", (16, p.in_channels, 64, 64))).cuda().repeat(B // 16, 1, 1, 1).contiguous()
res = {}
for mode in ("f32", "bf16x3"):
    net.set_precision(mode)
    for _ in range(2): net.predict_pose(hm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): pose = net.predict_pose(hm)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    res[mode] = pose.cpu().numpy()
    print(json.dumps({"B": B, "mode": mode, "ms_per_step": round(dt * 1e3, 2), "frames_per_s": round(B / dt, 1)}), flush=True)
print(json.dumps({"B": B, "max_abs_bf16x3_vs_f32": float(np.abs(res["bf16x3"] - res["f32"]).max())}))
