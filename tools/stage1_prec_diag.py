import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from egotap_amd import hm_ops as H
from egotap_amd.synthetic import synth_input
from test_gpu_hm_train_step import _net
outs = {}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for mode in ("f32", "bf16x3", "bf16"):
    net, _ = _net("pos")
    net.train()
    net.set_precision(mode)
    left = torch.from_numpy(synth_input("tr_rgbL_pos", (B, 3, 256, 256), -2.0, 2.0)).cuda()
    right = torch.from_numpy(synth_input("tr_rgbR_pos", (B, 3, 256, 256), -2.0, 2.0)).cuda()
    gt = torch.from_numpy(synth_input("tr_gt_pos", (B, 30, 64, 64), 0.0, 1.0)).cuda()
    pred = net(left, right)
    loss, dpred = H.mse(pred.detach().contiguous(), gt, None, 1.0)
    pred.backward(dpred)
    torch.cuda.synchronize()
    outs[mode] = (pred.detach().clone(), float(loss), {k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None})
for mode in ("bf16x3", "bf16"):
    p0, p1 = outs["f32"][0], outs[mode][0]
    print(mode, "pred rel L2", float((p1 - p0).norm() / p0.norm()), "max abs", float((p1 - p0).abs().max()), "max |p|", float(p0.abs().max()), "loss", outs[mode][1], outs["f32"][1])
    rows = []
    for k, g in outs["f32"][2].items():
        a, b = outs[mode][2][k].double().flatten(), g.double().flatten()
        if float(b.norm()) < 1e-9: continue
        rows.append((float((a - b).norm() / b.norm()), float(a @ b / (a.norm() * b.norm())), k))
    rows.sort(reverse=True)
    print("  worst rel:", [(round(r, 4), round(c, 4), k[-40:]) for r, c, k in rows[:4]])
    print("  median rel:", round(float(np.median([r for r, _, _ in rows])), 4), "min cos", round(min(c for _, c, _ in rows), 4))
