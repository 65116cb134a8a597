import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from gpu_util import lift_net
from egotap_amd.synthetic import synth_input
net, sd, p = lift_net("UnrealEgo")
hm = torch.from_numpy(synth_input("hm_lat", (1, p.in_channels, 64, 64))).cuda()
for _ in range(5): net.predict_pose(hm)
torch.cuda.synchronize()
for _ in range(10): net.predict_pose(hm)
torch.cuda.synchronize()
