#!/usr/bin/env python3
"""GPU probe: the bf16-storage weight-gradient GEMM (gemm_tn_bf16s_kernel) at the ViT shapes of the training step, HIP-event timed.
Run once per library build (EGOTAP_LIB) in ONE gpurun call to A/B a kernel change.  usage: python tools/tn_ab_probe.py [B = 1024]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from egotap_amd import bf16s  # noqa: E402


def timed(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
M = B * 576
tot = 0.0
for name, N, K in (("dW_qkv", 3072, 1024), ("dW_o", 1024, 1024), ("dW_up", 4096, 1024), ("dW_dn", 1024, 4096)):
    x = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
    dy = (torch.rand(M, N, device="cuda") - 0.5).bfloat16()
    dw = torch.empty((N, K), device="cuda")
    ms = sorted(timed(lambda: bf16s.gemm_tn(dy, x, dw)) for _ in range(3))[1]
    tot += ms
    print(json.dumps({"lib": os.environ.get("EGOTAP_LIB", "shipped").split("/")[-1], "role": name, "M": M, "N": N, "K": K, "ms": round(ms, 3),
                      "tf": round(2.0 * M * N * K / ms / 1e9, 1), "checksum": float(dw.double().abs().sum())}), flush=True)
    del x, dy, dw
print(json.dumps({"lib": os.environ.get("EGOTAP_LIB", "shipped").split("/")[-1], "sum_ms": round(tot, 3)}))
