#!/usr/bin/env python3
"""GPU probe: the EgoCap lifting head at 128 x 128 heatmaps (BASELINE config 5 geometry: 2304 tokens, fc1 K = 65536 / 32768) forward in a
given arithmetic at batch B (for rocprofv3 --stats).  usage: head_c5_probe.py B mode"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gpu_util import lift_net
from egotap_amd.synthetic import synth_input
B, mode = int(sys.argv[1]), sys.argv[2]
net, sd_np, p = lift_net("EgoCap", 128)
hm = torch.from_numpy(synth_input("hm_c5p", (4, p.in_channels, 128, 128))).cuda().repeat((B + 3) // 4, 1, 1, 1)[:B].contiguous()
net.set_precision(mode)
for _ in range(2):
    net.predict_pose(hm)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    net.predict_pose(hm)
torch.cuda.synchronize()
print(f"EgoCap-128 head B={B} {mode}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per forward")
