#!/usr/bin/env python3
"""GPU probe: speed of the bf16-storage GEMMs (gemm_bf16s_kernel NT, gemm_tn_bf16s_kernel TN) at the ViT shapes of the training step,
random operands, HIP-event timed; next to the round-1 kernel (egotap_linear_bf16_dma: bf16 operands, fp32 output)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egotap_amd import bf16s, lib


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    M = B * 576
    for N, K in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
        x = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
        w = ((torch.rand(N, K, device="cuda") - 0.5) * 0.1).bfloat16()
        b = torch.zeros(N, device="cuda")
        row = {"M": M, "N": N, "K": K}
        out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        ms = timed(lambda: bf16s.gemm_nt(x, w, b, out=out))
        row["nt_bf16out_ms"], row["nt_bf16out_tf"] = round(ms, 3), round(2.0 * M * N * K / ms / 1e9, 1)
        r = torch.zeros((M, N), device="cuda")
        ms = timed(lambda: bf16s.gemm_nt(x, w, b, epi="residual", aux=r, out=r))
        row["nt_res_f32_ms"], row["nt_res_f32_tf"] = round(ms, 3), round(2.0 * M * N * K / ms / 1e9, 1)
        if os.environ.get("PROBE_R1"):
            ms = timed(lambda: lib.linear_bf16_dma(x, w, b))
            row["r1_dma_f32out_ms"], row["r1_dma_f32out_tf"] = round(ms, 3), round(2.0 * M * N * K / ms / 1e9, 1)
        if N == 4096:
            ms = timed(lambda: bf16s.gemm_nt(x, w, b, epi="gelu_save"))
            row["nt_gelu_save_ms"], row["nt_gelu_save_tf"] = round(ms, 3), round(2.0 * M * N * K / ms / 1e9, 1)
            z = (torch.rand(M, N, device="cuda") - 0.5).bfloat16()
            ms = timed(lambda: bf16s.gemm_nt(x, w, None, epi="gelu_grad", aux=z))
            row["nt_gelu_grad_ms"], row["nt_gelu_grad_tf"] = round(ms, 3), round(2.0 * M * N * K / ms / 1e9, 1)
            del z
        if hasattr(bf16s, "gemm_tn"):
            dy = (torch.rand(M, N, device="cuda") - 0.5).bfloat16()
            dw = torch.empty((N, K), device="cuda")
            ms = timed(lambda: bf16s.gemm_tn(dy, x, dw))
            row["tn_ms"], row["tn_tf"] = round(ms, 3), round(2.0 * M * N * K / ms / 1e9, 1)
        print(json.dumps(row), flush=True)
        del x, w, out, r


if __name__ == "__main__":
    main()
