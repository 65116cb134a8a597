#!/usr/bin/env python3
"""GPU probe: bf16 attention forward / backward (egotap_bf16_attention_fwd / _bwd) at the training shapes, generation 1 (round 2's
register-staged kernels) against generation 2 (attention_bf16s2.h), interleaved in ONE process (cdna_hip_programming.md rule 24),
random gaussian operands.  usage: python tools/attn_probe.py [B] [N]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egotap_amd import bf16s, lib


def timed(fn, reps=5):
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
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 576
    heads, dh = 8, 128
    L = lib.load()
    M = B * N
    qkv = (torch.randn(M, 3 * heads * dh, device="cuda") * 0.5).bfloat16()
    dctx = (torch.randn(M, heads * dh, device="cuda") * 0.1).bfloat16()
    ctx, lse = bf16s.attention_fwd(qkv, B, N, heads)
    fl = 4.0 * B * heads * N * N * dh
    rows = {}
    outs = {}
    for rnd in range(3):
        for gen in (1, 2):
            lib.check(L.egotap_debug_attention_gen(gen))
            tf = timed(lambda: bf16s.attention_fwd(qkv, B, N, heads))
            tb = timed(lambda: bf16s.attention_bwd(qkv, ctx, dctx, lse, B, N, heads))
            rows.setdefault(gen, []).append((tf, tb))
            outs[gen] = bf16s.attention_bwd(qkv, ctx, dctx, lse, B, N, heads).float()
    for gen, r in rows.items():
        tf = sorted(x[0] for x in r)[len(r) // 2]
        tb = sorted(x[1] for x in r)[len(r) // 2]
        print(json.dumps({"gen": gen, "B": B, "N": N, "fwd_ms": round(tf, 3), "fwd_tf": round(fl / tf / 1e9, 1), "bwd_ms(dq+dkv)": round(tb, 3),
                          "bwd_tf_7products": round(3.5 * fl / tb / 1e9, 1)}))
    d = (outs[1] - outs[2]).abs().max().item()
    print("max |dqkv gen1 - gen2| =", d, " (scale", outs[1].abs().max().item(), ")")
    lib.check(L.egotap_debug_attention_gen(3))


if __name__ == "__main__":
    main()
