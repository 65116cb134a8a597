"""Development probe: the two heatmap estimators' forward in a given arithmetic, for rocprofv3.  usage: hm_bf16_probe.py B hm_size mode"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
from egotap_amd.synthetic import synth_input  # noqa: E402
from gpu_util import hm_net  # noqa: E402

B, hm, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
preset = "EgoCap" if hm == 128 else "UnrealEgo"
S = 4 * hm
l = torch.from_numpy(synth_input("probe_l", (4, 3, S, S), -2.0, 2.0)).cuda().repeat((B + 3) // 4, 1, 1, 1)[:B].contiguous()
r = torch.from_numpy(synth_input("probe_r", (4, 3, S, S), -2.0, 2.0)).cuda().repeat((B + 3) // 4, 1, 1, 1)[:B].contiguous()
nets = [hm_net(w, preset=preset, hm=hm)[0] for w in ("pos", "rot")]
for n in nets:
    n.set_precision(mode)
for _ in range(2):
    for n in nets:
        n(l, r)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    for n in nets:
        n(l, r)
torch.cuda.synchronize()
print(f"B={B} hm={hm} {mode}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per stereo batch (both estimators)")
