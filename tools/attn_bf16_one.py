#!/usr/bin/env python3
"""bf16-storage attention forward + backward a few times at B frames (for rocprofv3 --pmc passes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egotap_amd import bf16s
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
qkv = ((torch.rand(B * 576, 3072, device="cuda") - 0.5) * 4).bfloat16()
dctx = (torch.rand(B * 576, 1024, device="cuda") - 0.5).bfloat16()
for _ in range(4):
    ctx, lse = bf16s.attention_fwd(qkv, B, 576, 8)
    bf16s.attention_bwd(qkv, ctx, dctx, lse, B, 576, 8)
torch.cuda.synchronize()
