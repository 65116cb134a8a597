"""Development probe: per-role GEMM kernel and time of one lifting-head forward at batch B (the library's timing hook), for an A/B of
routing decisions across EGOTAP_LIB builds.  usage: role_times_probe.py B [mode]"""
import ctypes as C
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
from egotap_amd import lib  # noqa: E402
from egotap_amd.synthetic import synth_input  # noqa: E402
from gpu_util import lift_net  # noqa: E402

B = int(sys.argv[1])
mode = sys.argv[2] if len(sys.argv) > 2 else "f32"
net, sd, p = lift_net("UnrealEgo")
net.set_precision(mode)
hm = torch.from_numpy(synth_input("hm_lat", (B, p.in_channels, 64, 64))).cuda()
for _ in range(3):
    net.predict_pose(hm)
L, h = lib.load(), net._ensure_handle()
lib.check(L.egotap_timing_enable(h, 1))
for _ in range(10):
    net.predict_pose(hm)
torch.cuda.synchronize()
n, ms, fl = C.c_int(), C.c_double(), C.c_double()
lib.check(L.egotap_timing_read(h, C.byref(n), C.byref(ms), C.byref(fl)))
det = json.loads(L.egotap_timing_detail(h).decode())
lib.check(L.egotap_timing_enable(h, 0))
tot = 0.0
for d in det:
    us = d["ms"] / d["launches"] * 1e3
    tot += d["ms"] / 10 * 1e3
    print(f"{d['role']:12s} {us:8.1f} us  {d['kernel'][:90]}")
print(f"GEMM total per forward: {tot:.1f} us")
