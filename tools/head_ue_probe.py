#!/usr/bin/env python3
"""GPU probe: the UnrealEgo lifting head forward at batch B in a given arithmetic (for rocprofv3 --stats).  usage: head_ue_probe.py B mode"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from gpu_util import lift_net
from egotap_amd.synthetic import synth_input
B, mode = int(sys.argv[1]), sys.argv[2]
net, sd_np, p = lift_net("UnrealEgo")
hm = torch.from_numpy(synth_input("hm_uep", (4, p.in_channels, 64, 64))).cuda().repeat((B + 3) // 4, 1, 1, 1)[:B].contiguous()
net.set_precision(mode)
for _ in range(2):
    net.predict_pose(hm)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    net.predict_pose(hm)
torch.cuda.synchronize()
print(f"UnrealEgo head B={B} {mode}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per forward = {B * 5 / (time.perf_counter() - t0):.0f} frames/s")
