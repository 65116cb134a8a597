#!/usr/bin/env python3
"""GPU probe: gemm_bf16_dma_kernel (egotap_linear_f32 tile 17) against torch on bf16-rounded operands, and its speed next to
the register-staged bf16 kernel (tile 16) and the exact-fp32 kernel (tile 12)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egotap_amd import lib

def check(M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.rand(M, K, device="cuda", generator=g) - 0.5
    w = (torch.rand(N, K, device="cuda", generator=g) - 0.5) * 0.1
    b = torch.rand(N, device="cuda", generator=g)
    y = lib.linear(x, w, b, tile=17)
    ref = (x.bfloat16().float() @ w.bfloat16().float().T) + b
    err = (y - ref).abs().max().item()
    print(json.dumps({"M": M, "N": N, "K": K, "max_abs_err_vs_bf16_operands_f32_acc": err, "ref_max": ref.abs().max().item()}), flush=True)
    assert err < 2e-4 * max(1.0, K / 1024), err

def speed(M, N, K, tiles=(12, 16, 17, 18)):
    x = torch.rand(M, K, device="cuda") - 0.5
    w = (torch.rand(N, K, device="cuda") - 0.5) * 0.1
    b = torch.zeros(N, device="cuda")
    row = {"M": M, "N": N, "K": K}
    for t in tiles:
        for _ in range(2):
            lib.linear(x, w, b, tile=t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.linear(x, w, b, tile=t)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        row[f"tile{t}_ms"] = round(ms, 3)
        row[f"tile{t}_tf"] = round(2.0 * M * N * K / ms / 1e9, 1)
    print(json.dumps(row), flush=True)

def check_f32(M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.rand(M, K, device="cuda", generator=g) - 0.5
    w = (torch.rand(N, K, device="cuda", generator=g) - 0.5) * 0.1
    b = torch.rand(N, device="cuda", generator=g)
    same = torch.equal(lib.linear(x, w, b, tile=19), lib.linear(x, w, b, tile=12))
    print(json.dumps({"M": M, "N": N, "K": K, "f32_dma_bit_identical_to_persist": same}), flush=True)
    assert same

if __name__ == "__main__":
    for shp in ((256, 256, 32), (256, 256, 128), (300, 512, 1024), (1000, 1024, 4096), (8192, 3072, 1024)):
        check(*shp)
    if len(sys.argv) > 1 and sys.argv[1] == "f32":
        for shp in ((256, 256, 16), (300, 512, 48), (1153, 768, 1024), (5000, 256, 32)):
            check_f32(*shp)
        for shp in ((147456, 1024, 4096), (147456, 3072, 1024), (147456, 1024, 1024), (147456, 4096, 1024)):
            speed(*shp, tiles=(12, 19))
        sys.exit(0)
    speed(147456, 1024, 4096)
    speed(147456, 3072, 1024)
    speed(147456, 1024, 1024)
    speed(147456, 4096, 1024)
