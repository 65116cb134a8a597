"""Development probe: fp32 evaluation forward of a Bottleneck estimator (resnet50 / resnet101) at batch B.  usage: hm_bottleneck_probe.py model B"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
from egotap_amd.synthetic import synth_input  # noqa: E402
from gpu_util import hm_net  # noqa: E402

model, B = sys.argv[1], int(sys.argv[2])
l = torch.from_numpy(synth_input("probe_l", (4, 3, 256, 256), -2.0, 2.0)).cuda().repeat((B + 3) // 4, 1, 1, 1)[:B].contiguous()
r = torch.from_numpy(synth_input("probe_r", (4, 3, 256, 256), -2.0, 2.0)).cuda().repeat((B + 3) // 4, 1, 1, 1)[:B].contiguous()
net = hm_net("pos", model_name=model)[0]
net(l, r)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    net(l, r)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"{model} position estimator, B={B}: {dt * 1e3:.1f} ms per stereo batch = {B / dt:.0f} frames/s, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
