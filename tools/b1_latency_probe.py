"""Development probe for rocprofv3 --kernel-trace: 5 warm-up + 10 forwards of the lifting head at batch B in a given arithmetic.
usage: b1_latency_probe.py [B] [mode]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
from egotap_amd.synthetic import synth_input  # noqa: E402
from gpu_util import lift_net  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mode = sys.argv[2] if len(sys.argv) > 2 else "f32"
net, sd, p = lift_net("UnrealEgo")
net.set_precision(mode)
hm = torch.from_numpy(synth_input("hm_lat", (B, p.in_channels, 64, 64))).cuda()
for _ in range(5):
    net.predict_pose(hm)
torch.cuda.synchronize()
for _ in range(10):
    net.predict_pose(hm)
torch.cuda.synchronize()
