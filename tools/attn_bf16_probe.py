#!/usr/bin/env python3
"""GPU probe: bf16-storage attention forward / backward alone at B frames, HIP-event timed"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egotap_amd import bf16s
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
qkv = ((torch.rand(B * 576, 3072, device="cuda") - 0.5) * 4).bfloat16()
dctx = (torch.rand(B * 576, 1024, device="cuda") - 0.5).bfloat16()
ctx, lse = bf16s.attention_fwd(qkv, B, 576, 8)
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
f = timed(lambda: bf16s.attention_fwd(qkv, B, 576, 8))
b = timed(lambda: bf16s.attention_bwd(qkv, ctx, dctx, lse, B, 576, 8))
fl = 4.0 * B * 8 * 576 * 576 * 128
print(json.dumps({"B": B, "fwd_ms": round(f, 3), "fwd_tf": round(fl / f / 1e9, 1), "bwd_ms": round(b, 3), "bwd_tf_5products": round(2.5 * fl / b / 1e9, 1)}))
