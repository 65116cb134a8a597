import os, sys
sys.path.insert(0, os.getcwd())
import torch
from egotap_amd import bf16s
from egotap_amd.train_ops import _scratch
B = 1024; M = B * 576
for (N, K) in ((1024, 1024), (1024, 4096)):
    x = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
    dy = (torch.rand(M, N, device="cuda") - 0.5).bfloat16()
    dw = torch.empty((N, K), device="cuda")
    for rep in range(3):
        bf16s.gemm_tn(dy, x, dw)
        torch.cuda.synchronize()
        tiles = (N // 256) * (K // 256); splits = max(1, min((256 + tiles - 1) // tiles, (M + 255) // 256))
        off = (splits * N * K * 4 + 255) & ~255
        c = _scratch.buf[off: off + 8 * splits * 4].view(torch.int32).cpu()
        steps = M // splits // 32
        print(N, K, "splits", splits, "sync points", (steps - 1) // 128, "counters min/max", int(c.min()), int(c.max()), "expected", (256 // 8 // max(1, 32 // tiles) if False else None), c[:8].tolist())
