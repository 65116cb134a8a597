#!/usr/bin/env python3
"""debug: the backward operators of the six FC blocks fed with the float64 oracle's own tensors of a real step (z, y, dy of every
block), each against the oracle's result for that operator.  GPU box only."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from egotap_amd import spec, networks, train_ops as T
from egotap_amd.synthetic import synth_input, synth_state_dict
from egotap_amd.options import preset_defaults
from oracle import lift_ref as O

preset = sys.argv[1] if len(sys.argv) > 1 else "UnrealEgo"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
p = spec.lift_preset(preset)
sd_np = synth_state_dict(spec.lift_state_spec(p))
net = networks.EgoTAPAutoEncoder(preset_defaults(preset), input_channel_scale=2)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
net = net.cuda().train()
h = net._ensure_handle()
hm = torch.from_numpy(synth_input("wrap_hm_ec_step", (B, p.in_channels, 64, 64))).double()
gt = torch.from_numpy(synth_input("wrap_gt_ec_step", (B, p.out_joints, 3), -20.0, 20.0)).double()
sd = O.to_torch_sd(sd_np, torch.float64)

rec = {}
orig = O.fc_block


def spy(x, sd_, prefix, training=False, **kw):
    out = orig(x, sd_, prefix, training, **kw)
    y = out[0] if training else out
    if y.requires_grad:
        y.retain_grad()
        if x.requires_grad:
            x.retain_grad()
    rec[prefix] = (x, y)
    return out


O.fc_block = spy
leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
full = dict(sd); full.update(leaves)
pose, _ = O.lift_forward_train(hm, full, p)
loss = O.loss_mpjpe(pose, gt) * 0.1 + O.loss_cos_sim(pose, gt, p) * (-0.01) * 0.1
loss.backward()


def rel(a, b):
    b = b.double().cpu()
    return float((a.double().cpu() - b).abs().max() / (b.norm() / np.sqrt(b.numel())).clamp_min(1e-30))


cu = lambda t: t.detach().float().cuda().contiguous()
for enc in ("rot_heatmap_encoder", "pos_heatmap_encoder"):
    for name in ("fc3", "fc2", "fc1"):
        pre = f"{enc}.{name}"
        x, y = rec[pre]
        W, b, g, beta = (full[pre + s].detach() for s in (".fc.weight", ".fc.bias", ".bn.weight", ".bn.bias"))
        xd = x.detach()
        dy = y.grad
        z = (xd @ W.T + b).requires_grad_(True)
        mean = z.mean(0); var = ((z - mean) ** 2).mean(0)
        y2 = torch.nn.functional.leaky_relu((z - mean) / torch.sqrt(var + 1e-5) * g + beta, 0.2)
        (dz,) = torch.autograd.grad(y2, z, dy)
        R, N = z.shape
        K = xd.shape[1]
        # forward statistics on the GPU from the oracle's z
        zc = cu(z)
        yg, mg, rg = T.bn_lrelu_fwd(zc, cu(g), cu(beta), torch.zeros(N, device="cuda"), torch.ones(N, device="cuda"))
        dgm, dbt = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
        dzg = T.bn_lrelu_bwd(zc, yg, cu(dy), cu(g), mg, rg, dgm, dbt)
        row = [f"{pre} R={R} N={N} K={K}", f"y {rel(yg, y2):.1e}", f"dz {rel(dzg, dz):.1e}", f"dgamma {rel(dgm, full[pre + '.bn.weight'].grad):.1e}",
               f"dbeta {rel(dbt, full[pre + '.bn.bias'].grad):.1e}"]
        # the same with the oracle's dz: weight gradient (plain loader needs the gathered rows: fc2 / fc3 only), input gradient
        dzc = cu(dz)
        if name != "fc1":
            dw = torch.empty(N, K, device="cuda")
            T.gemm_tn(h, dzc, cu(xd), dw, R, N, K)
            row.append(f"dW {rel(dw, full[pre + '.fc.weight'].grad):.1e}")
            wt = T.transpose(cu(W))
            dx = T.gemm_nt(h, dzc, wt, None, R, K, N, epi=T.TE_NONE)
            row.append(f"dx {rel(dx, x.grad):.1e}")
        cs = torch.empty(N, device="cuda")
        T.colsum(dzc, cs, R, N)
        row.append(f"colsum {float((cs.double().cpu() - dz.sum(0)).abs().max() / dz.abs().sum(0).max()):.1e}")
        # how sharp the block is: smallest batch variance, largest rstd * gamma
        row.append(f"min var {float(var.min()):.2e}  max |g|*rstd {float((g.abs() / torch.sqrt(var + 1e-5)).max()):.1f}")
        print("  ".join(row), flush=True)

# ---- the real step, composed operator by operator (training.LiftTrainFn), every BatchNorm backward's inputs and output recorded
from egotap_amd import training
from egotap_amd.training import PoseLossFn
calls = []
real = T.bn_lrelu_bwd


def spy_bwd(z, y, dy, gamma, mean, rstd, dgamma, dbeta, accumulate=False):
    dz = real(z, y, dy, gamma, mean, rstd, dgamma, dbeta, accumulate)
    calls.append(dict(z=z.clone(), y=y.clone(), dy=dy.clone(), dz=dz.clone(), mean=mean.clone(), rstd=rstd.clone()))
    return dz


for one_call in (False, True):
    net.zero_grad(set_to_none=True)
    net.one_call_training = one_call
    calls.clear()
    training.T.bn_lrelu_bwd = spy_bwd
    PoseLossFn.apply(net, net(hm.float().cuda())[0], gt.float().cuda(), 0.1, -0.01).sum().backward()
    torch.cuda.synchronize()
    training.T.bn_lrelu_bwd = real
    print("one_call", one_call, "recorded", len(calls))
    order = [f"{e}.{n}" for e in ("rot_heatmap_encoder", "pos_heatmap_encoder") for n in ("fc3", "fc2", "fc1")]
    for pre, c in zip(order, calls):
        x, y = rec[pre]
        W, b, g, beta = (full[pre + s].detach() for s in (".fc.weight", ".fc.bias", ".bn.weight", ".bn.bias"))
        z = (x.detach() @ W.T + b).requires_grad_(True)
        mean = z.mean(0); var = ((z - mean) ** 2).mean(0)
        y2 = torch.nn.functional.leaky_relu((z - mean) / torch.sqrt(var + 1e-5) * g + beta, 0.2)
        (dz,) = torch.autograd.grad(y2, z, y.grad)
        print(f"  {pre}: z {rel(c['z'], z.detach()):.1e}  y {rel(c['y'], y2.detach()):.1e}  mean {rel(c['mean'], mean.detach()):.1e}  rstd {rel(c['rstd'], 1 / torch.sqrt(var.detach() + 1e-5)):.1e}"
              f"  dy {rel(c['dy'], y.grad):.1e}  dz {rel(c['dz'], dz):.1e}", flush=True)
    for k in ("pos_heatmap_encoder.fc1.fc.weight", "pos_heatmap_encoder.fc1.bn.bias", "pos_heatmap_encoder.fc2.fc.weight", "pos_heatmap_encoder.fc2.bn.bias", "rot_heatmap_encoder.fc2.fc.weight"):
        print(f"  grad {k}: {rel(dict(net.named_parameters())[k].grad, full[k].grad):.1e}")
