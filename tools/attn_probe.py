#!/usr/bin/env python3
"""GPU probe: the fp32 attention forward alone at the headline shape (B x 8 heads x 576 tokens x 128), HIP-event timed"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egotap_amd import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
qkv = (torch.rand(B * 576, 3072, device="cuda") - 0.5) * 4
for _ in range(3):
    lib.attention(qkv, B, 576, 8)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.attention(qkv, B, 576, 8)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 5)
ms = sorted(ts)[len(ts) // 2]
fl = 4.0 * B * 8 * 576 * 576 * 128
print(json.dumps({"B": B, "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1), "frac_of_f32_peak": round(fl / ms / 1e9 / 157.3, 4)}))
