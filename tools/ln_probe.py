#!/usr/bin/env python3
"""GPU probe: the bf16-storage LayerNorm forward / backward alone at config 3's row count, HIP-event timed, with checksums for an A/B across
EGOTAP_LIB builds.  usage: python tools/ln_probe.py [rows]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egotap_amd import bf16s
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 589824
torch.manual_seed(0)
x = torch.randn(rows, 1024, device="cuda")
dy = (torch.randn(rows, 1024, device="cuda") * 0.1).bfloat16()
dres = torch.randn(rows, 1024, device="cuda") * 0.1
g, b = torch.rand(1024, device="cuda") + 0.5, torch.randn(1024, device="cuda")
dg, db, dc = (torch.zeros(1024, device="cuda") for _ in range(3))
y, mean, rstd = bf16s.layernorm_fwd(x, g, b)
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tf = sorted(timed(lambda: bf16s.layernorm_fwd(x, g, b)) for _ in range(3))[1]
tb = sorted(timed(lambda: bf16s.layernorm_bwd(x, dy, g, mean, rstd, dg, db, dres=dres, dcolsum=dc)) for _ in range(3))[1]
dx, dxb = bf16s.layernorm_bwd(x, dy, g, mean, rstd, dg, db, dres=dres, dcolsum=dc)
print(json.dumps({"lib": os.environ.get("EGOTAP_LIB", "shipped"), "rows": rows, "ln_fwd_ms": round(tf, 4), "fwd_TBps": round(rows * 1024 * 6 / tf / 1e9, 2),
                  "ln_bwd_ms": round(tb, 4), "bwd_TBps": round(rows * 1024 * 16 / tb / 1e9, 2),
                  "checksum": [float(y.float().abs().sum()), float(dx.abs().sum()), float(dxb.float().abs().sum()), float(dg.abs().sum())]}))
