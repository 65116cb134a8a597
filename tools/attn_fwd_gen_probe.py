#!/usr/bin/env python3
"""GPU probe: the bf16-storage attention forward generations (egotap_debug_attention_gen 1 / 2 / 3) at B frames, HIP-event timed, and
the bitwise relation between their outputs"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egotap_amd import bf16s, lib
L = lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 576
qkv = ((torch.rand(B * N, 3072, device="cuda") - 0.5) * 4).bfloat16()
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
fl = 4.0 * B * 8 * N * N * 128
outs = {}
for rep in range(2):
    for gen in (2, 34, 33, 32):
        lib.check(L.egotap_debug_attention_gen(gen))
        ms = timed(lambda: bf16s.attention_fwd(qkv, B, N, 8))
        ctx, lse = bf16s.attention_fwd(qkv, B, N, 8)
        outs[gen] = (ctx.clone(), lse.clone())
        print(json.dumps({"gen": gen, "B": B, "N": N, "fwd_ms": round(ms, 3), "fwd_tf": round(fl / ms / 1e9, 1)}), flush=True)
lib.check(L.egotap_debug_attention_gen(3))
for g in (34, 33, 32):
    d = (outs[g][0].float() - outs[2][0].float()).abs().max().item()
    dl = (outs[g][1] - outs[2][1]).abs().max().item()
    print(f"gen {g} vs gen 2: max |ctx diff| {d:.3e} (equal: {torch.equal(outs[g][0], outs[2][0])}), max |lse diff| {dl:.3e}")
