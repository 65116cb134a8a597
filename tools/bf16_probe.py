#!/usr/bin/env python3
"""GPU probe: bf16x3 / bf16 MFMA GEMM (tiles 13 / 14) against the exact-fp32 kernel (tile 12): error vs float64 and TFLOP/s."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egotap_amd import lib

def err(M, N, K, tile):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    y = lib.linear(x, w, b, tile=tile)
    ref = x.double() @ w.double().t() + b.double()
    d = (y.double() - ref).abs()
    return float(d.max()), float(d.pow(2).mean().sqrt()), float(ref.abs().mean())

def bench(M, N, K, tile, reps=5):
    x = (torch.rand(M, K, device="cuda") - 0.5)
    w = (torch.rand(N, K, device="cuda") - 0.5) * 0.1
    b = torch.zeros(N, device="cuda")
    lib.linear(x, w, b, tile=tile)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        lib.linear(x, w, b, tile=tile)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    return 2.0 * M * N * K / (ms * 1e-3) / 1e12, ms

if __name__ == "__main__":
    L = lib.load()
    tiles = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [12, 13, 14]
    for (m, n, k) in [(1000, 512, 1024), (4096, 1024, 4096)]:
        for tile in tiles:
            mx, rms, mag = err(m, n, k, tile)
            print(json.dumps({"check": [m, n, k], "tile": L.egotap_gemm_tile_name(tile).decode(), "max_err": mx, "rms_err": rms, "mean_abs_ref": mag}), flush=True)
    M = 147456
    for (m, n, k) in [(M, 3072, 1024), (M, 1024, 1024), (M, 4096, 1024), (M, 1024, 4096), (7680, 2048, 16384)]:
        for tile in tiles:
            tf, ms = bench(m, n, k, tile)
            print(json.dumps({"M": m, "N": n, "K": k, "tile": L.egotap_gemm_tile_name(tile).decode(), "tflops": round(tf, 1), "ms": round(ms, 3)}), flush=True)
