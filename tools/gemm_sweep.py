#!/usr/bin/env python3
"""GPU microbench: fp32 MFMA GEMM tile shapes on the ViT shapes of the B=256 lifting head."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egotap_amd import lib

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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 147456
    shapes = [(M, 3072, 1024), (M, 1024, 1024), (M, 4096, 1024), (M, 1024, 4096), (7680, 2048, 16384)]
    L = lib.load()
    for (m, n, k) in shapes:
        for tile in (12,):
            tf, ms = bench(m, n, k, tile)
            print(json.dumps({"M": m, "N": n, "K": k, "tile": L.egotap_gemm_tile_name(tile).decode(), "tflops": round(tf, 1), "ms": round(ms, 3)}), flush=True)
