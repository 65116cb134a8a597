#!/usr/bin/env python3
"""GPU probe: lifting-head latency at small batch (serving), fp32 and bf16x3"""
import sys, os, json, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from egotap_amd.synthetic import synth_input
from gpu_util import lift_net
net, _, p = lift_net("UnrealEgo")
BS = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else (1, 2, 4, 8, 16, 32, 64)
MODES = sys.argv[2].split(",") if len(sys.argv) > 2 else ("f32", "bf16x3")
for B in BS:
    hm = torch.from_numpy(synth_input("hm_lat", (B, p.in_channels, 64, 64))).cuda()
    row = {"B": B}
    for mode in MODES:
        net.set_precision(mode)
        for _ in range(3): net.predict_pose(hm)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): net.predict_pose(hm)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        row[mode + "_ms"] = round(dt * 1e3, 3)
    print(json.dumps(row), flush=True)
