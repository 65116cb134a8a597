#!/usr/bin/env python3
"""GPU probe: the fp32 attention forward (egotap_train_attention_fwd) at B frames, HIP-event timed, with a checksum of its output.
Run once per library (EGOTAP_LIB=<variant>) in one gpurun call for an A/B.  usage: attn_f32_probe.py [B] [N]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egotap_amd import train_ops as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 576
torch.manual_seed(0)
qkv = (torch.rand(B * N, 3072, device="cuda") - 0.5) * 4
def run():
    return T.attention_fwd(qkv, B, N, 8, "f32")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ctx, lse = run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(json.dumps({"lib": os.environ.get("EGOTAP_LIB", "default"), "B": B, "N": N, "ms": round(ms, 4), "tf": round(4.0 * B * 8 * N * N * 128 / ms / 1e9, 1),
                  "ctx_sum": float(ctx.double().sum()), "ctx_abs": float(ctx.double().abs().sum()), "lse_sum": float(lse.double().sum())}))
