#!/usr/bin/env python3
"""GPU probe: the bf16-storage NT GEMM on 32-deep (csrc/gemm_bf16s.h) against 64-deep K-tiles (csrc/gemm_bf16s64.h) at the ViT shapes of
the training step, every epilogue the step uses -- interleaved rounds in ONE process (egotap_debug_gemm_bk), random operands, HIP-event
timed, median of the rounds.  usage: python tools/gemm_bk_probe.py [B = 1024] [rounds = 3]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from egotap_amd import bf16s, lib  # noqa: E402


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


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    L = lib.load()
    M = B * 576
    cases = [("qkv", 3072, 1024, "bf16"), ("attn_out", 1024, 1024, "residual"), ("dctx", 1024, 1024, "bf16"), ("mlp_up", 4096, 1024, "gelu_save"),
             ("mlp_down", 1024, 4096, "residual"), ("dhid", 4096, 1024, "gelu_grad"), ("dy2", 1024, 4096, "bf16"), ("dy1", 1024, 3072, "bf16")]
    tot = {32: 0.0, 64: 0.0}
    for name, N, K, epi in cases:
        x = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
        w = ((torch.rand(N, K, device="cuda") - 0.5) * 0.1).bfloat16()
        b = torch.rand(N, device="cuda")
        kw = {}
        if epi == "residual":
            r = torch.zeros((M, N), device="cuda")
            kw = dict(aux=r, out=r)
        elif epi == "gelu_grad":
            kw = dict(aux=(torch.rand(M, N, device="cuda") - 0.5).bfloat16(), out=torch.empty((M, N), dtype=torch.bfloat16, device="cuda"),
                      colsum_out=torch.empty(N, device="cuda"))
        elif epi == "gelu_save":
            kw = dict(out=torch.empty((M, N), dtype=torch.bfloat16, device="cuda"), out1=torch.empty((M, N), dtype=torch.bfloat16, device="cuda"))
        else:
            kw = dict(out=torch.empty((M, N), dtype=torch.bfloat16, device="cuda"))
        bias = None if epi == "gelu_grad" else b
        ms = {32: [], 64: []}
        for _ in range(rounds):
            for bk in (32, 64):
                lib.check(L.egotap_debug_gemm_bk(bk))
                ms[bk].append(timed(lambda: bf16s.gemm_nt(x, w, bias, epi=epi, **kw)))
        lib.check(L.egotap_debug_gemm_bk(0))
        med = {bk: sorted(v)[len(v) // 2] for bk, v in ms.items()}
        fl = 2.0 * M * N * K
        for bk in med:
            tot[bk] += med[bk]
        print(json.dumps({"role": name, "M": M, "N": N, "K": K, "epi": epi, "bk32_ms": round(med[32], 3), "bk64_ms": round(med[64], 3),
                          "bk32_tf": round(fl / med[32] / 1e9, 1), "bk64_tf": round(fl / med[64] / 1e9, 1), "gain_pct": round(100 * (med[32] / med[64] - 1), 1)}), flush=True)
        del x, w, kw
        torch.cuda.empty_cache()
    print(json.dumps({"sum_ms_per_layer": {k: round(v, 2) for k, v in tot.items()}, "gain_pct": round(100 * (tot[32] / tot[64] - 1), 1)}))


if __name__ == "__main__":
    main()
