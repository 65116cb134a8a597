import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egotap_amd import lib
M, N, K, tile = 147456, 1024, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 12
x = torch.rand(M, K, device="cuda") - 0.5; w = (torch.rand(N, K, device="cuda") - 0.5) * 0.1; b = torch.zeros(N, device="cuda")
for _ in range(3):
    lib.linear(x, w, b, tile=tile)
torch.cuda.synchronize()
