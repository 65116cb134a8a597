#!/usr/bin/env python3
"""debug: where does the redone (pu_solo_kernel) recurrence differ from the chain?  GPU box only."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from egotap_amd import lib as _lib, networks, spec
from egotap_amd.options import preset_defaults
from egotap_amd.synthetic import synth_input, synth_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
p = spec.lift_preset("UnrealEgo")
net = networks.EgoTAPAutoEncoder(preset_defaults("UnrealEgo"), input_channel_scale=2)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(p)).items()})
net = net.cuda().eval()
L, h = _lib.load(), net._ensure_handle()
hm = torch.from_numpy(synth_input("hm_chain_fault", (B, p.in_channels, 64, 64))).cuda()
want = net.predict_pose(hm).clone()
hs_want = net.intermediate("skel_embed", B).clone().view(p.n_joints_hm, B, 512)
net.set_pu_chain(False)
step = net.predict_pose(hm).clone()
hs_step = net.intermediate("skel_embed", B).clone().view(p.n_joints_hm, B, 512)
print("per-step vs chain equal:", torch.equal(step, want), torch.equal(hs_step, hs_want))
net.set_pu_chain(True)
_lib.check(L.egotap_debug_pu_drop_workgroups(h, 1))
got = net.predict_pose(hm).clone()
torch.cuda.synchronize()
_lib.check(L.egotap_debug_pu_drop_workgroups(h, 0))
hs_got = net.intermediate("skel_embed", B).clone().view(p.n_joints_hm, B, 512)
print("status", net.pu_chain_status(), "pose equal", torch.equal(got, want), "max diff", float((got - want).abs().max()))
d = (hs_got - hs_want).abs()
print("HS1 max diff per step:", [f"{float(d[t].max()):.2e}" for t in range(p.n_joints_hm)])
t0 = next((t for t in range(p.n_joints_hm) if float(d[t].max()) > 0), None)
if t0 is not None:
    bad = (d[t0] > 0).nonzero()
    print("first bad step", t0, "count", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:20], "units min/max", int(bad[:, 1].min()), int(bad[:, 1].max()))
    print("sample", bad[:8].tolist(), hs_got[t0][bad[0, 0], bad[0, 1]].item(), hs_want[t0][bad[0, 0], bad[0, 1]].item())
