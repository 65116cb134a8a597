"""Development probe: B = 1 / B = 8 forward latency of the lifting head (median of 30, synchronised), for same-call A/B of two builds
(EGOTAP_LIB selects the library).  usage: latency_ab_probe.py [preset] [comma-separated batch sizes] [comma-separated modes]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
from egotap_amd.synthetic import synth_input  # noqa: E402
from gpu_util import lift_net  # noqa: E402

net, sd, p = lift_net(sys.argv[1] if len(sys.argv) > 1 else "UnrealEgo")
out = {}
BS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8, 16, 32]
MODES = sys.argv[3].split(",") if len(sys.argv) > 3 else ["f32", "bf16x3", "bf16"]
for B in BS:
    hm = torch.from_numpy(synth_input("hm_lat", (B, p.in_channels, p.hm_size, p.hm_size))).cuda()
    for mode in MODES:
        net.set_precision(mode)
        for _ in range(3):
            net.predict_pose(hm)
        ts = []
        for _ in range(30 if B <= 32 else 8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            net.predict_pose(hm)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        out[f"b{B}_{mode}"] = round(1e3 * ts[len(ts) // 2], 3)
print(out)
