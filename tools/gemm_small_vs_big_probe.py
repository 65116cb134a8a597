"""Development probe: the exact-fp32 GEMM of a ViT role at serving batch sizes on the 128 x 128 kernel (tile 1) and on the 256 x 256
LDS-DMA kernel (tile 19), to calibrate gemm_small's routing estimate (egotap_abi.hip).  usage: gemm_small_vs_big_probe.py"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from egotap_amd import lib  # noqa: E402


def t_of(fn, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for B in (6, 8, 10, 12, 16, 20, 24, 32, 40, 48, 64, 96):
    M = B * 576
    row = {"B": B}
    for name, N, K in (("qkv", 3072, 1024), ("out", 1024, 1024), ("up", 4096, 1024), ("down", 1024, 4096)):
        x = torch.rand(M, K, device="cuda") - 0.5
        w = (torch.rand(N, K, device="cuda") - 0.5) * 0.1
        b = torch.zeros(N, device="cuda")
        small = t_of(lambda: lib.linear(x, w, b, tile=1))
        big = t_of(lambda: lib.linear(x, w, b, tile=19))
        t128, t256 = ((M + 127) // 128) * (N // 128), ((M + 255) // 256) * (N // 256)
        row[name] = f"small {small:7.1f} big {big:7.1f} us  (tiles128/CU {t128 / 256:5.2f}, rounds256 {t256 / 256:4.2f})"
    print(row)
