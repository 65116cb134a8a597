import sys, os, json
sys.path.insert(0, "/root/repo")
import torch
from egotap_amd import lib
L = lib.load()
def bench(x, w, b, tile, reps=5):
    lib.linear(x, w, b, tile=tile); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps): lib.linear(x, w, b, tile=tile)
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps
M, N, K = 147456, 4096, 1024
b = torch.zeros(N, device="cuda")
for name, x, w in (("random", torch.rand(M, K, device="cuda") - 0.5, (torch.rand(N, K, device="cuda") - 0.5) * 0.1),
                   ("zeros", torch.zeros(M, K, device="cuda"), torch.zeros(N, K, device="cuda"))):
    for tile in (12, 15):
        ms = bench(x, w, b, tile)
        print(json.dumps({"data": name, "tile": L.egotap_gemm_tile_name(tile).decode(), "ms": round(ms, 3), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}), flush=True)
