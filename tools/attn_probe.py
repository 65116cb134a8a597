#!/usr/bin/env python3
"""GPU probe: the bf16-storage attention kernels (forward, dQ + dK/dV backward) at B frames of N tokens, HIP-event timed (median of
three rounds), with the run-to-run bit equality of their outputs.  usage: python tools/attn_probe.py [B] [N]
To A/B a kernel change build the variant into its own library (EGOTAP_LIB=<path> EGOTAP_CXXFLAGS=-D... python -m egotap_amd.build) and
run this probe once per library in ONE gpurun call, dumping outputs with --dump <file> to compare bits across the two runs."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from egotap_amd import bf16s  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B = int(args[0]) if len(args) > 0 else 1024
    N = int(args[1]) if len(args) > 1 else 576
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    heads, dh = 8, 128
    M = B * N
    torch.manual_seed(0)
    qkv = (torch.randn(M, 3 * heads * dh, device="cuda") * 0.5).bfloat16()
    dctx = (torch.randn(M, heads * dh, device="cuda") * 0.1).bfloat16()
    ctx, lse = bf16s.attention_fwd(qkv, B, N, heads)
    fl = 4.0 * B * heads * N * N * dh
    rows = []
    for _ in range(3):
        tf = timed(lambda: bf16s.attention_fwd(qkv, B, N, heads))
        tb = timed(lambda: bf16s.attention_bwd(qkv, ctx, dctx, lse, B, N, heads))
        rows.append((tf, tb))
    tf, tb = sorted(x[0] for x in rows)[1], sorted(x[1] for x in rows)[1]
    dq = bf16s.attention_bwd(qkv, ctx, dctx, lse, B, N, heads)
    dq2 = bf16s.attention_bwd(qkv, ctx, dctx, lse, B, N, heads)
    ctx2, lse2 = bf16s.attention_fwd(qkv, B, N, heads)
    print(json.dumps({"lib": os.environ.get("EGOTAP_LIB", "shipped"), "B": B, "N": N, "fwd_ms": round(tf, 3), "fwd_tf": round(fl / tf / 1e9, 1),
                      "bwd_ms(dq+dkv)": round(tb, 3), "bwd_tf_7products": round(3.5 * fl / tb / 1e9, 1),
                      "bit_reproducible": bool(torch.equal(dq, dq2) and torch.equal(ctx, ctx2) and torch.equal(lse, lse2)),
                      "checksum": [float(ctx.float().abs().sum()), float(lse.abs().sum()), float(dq.float().abs().sum())]}), flush=True)
    if dump:
        torch.save({"ctx": ctx.cpu(), "lse": lse.cpu(), "dqkv": dq.cpu()}, dump)


if __name__ == "__main__":
    main()
