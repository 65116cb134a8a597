#!/usr/bin/env python3
"""one bf16-storage GEMM launched a few times (for rocprofv3 --pmc passes): bf16s_one.py nt|res|tn B N K"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egotap_amd import bf16s

kind, B, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
M = B * 576
x = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
w = ((torch.rand(N, K, device="cuda") - 0.5) * 0.1).bfloat16()
b = torch.zeros(N, device="cuda")
for _ in range(4):
    if kind == "nt":
        bf16s.gemm_nt(x, w, b)
    elif kind == "res":
        r = torch.zeros((M, N), device="cuda")
        bf16s.gemm_nt(x, w, b, epi="residual", aux=r, out=r)
    else:
        dy = (torch.rand(M, N, device="cuda") - 0.5).bfloat16()
        bf16s.gemm_tn(dy, x, torch.empty((N, K), device="cuda"))
torch.cuda.synchronize()
