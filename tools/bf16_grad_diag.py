#!/usr/bin/env python3
"""GPU diagnostic: gradient agreement (cosine, relative L2) of one bf16 training step with the float64 oracle, per large tensor, for
the bf16-storage path and for the fp32-storage path (net.bf16_storage = False): bf16_grad_diag.py [preset] [hm] [B]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from test_gpu_configs import _model, _data, _oracle_step
preset = sys.argv[1] if len(sys.argv) > 1 else "UnrealEgo"
hm = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
m, p = _model(preset, hm, use_amp=True)
data, hmt, gt = _data(B, p, "diag", gt_range=1.0)
ref = _oracle_step(hmt, gt, p)
for storage in (True, False):
    m, p = _model(preset, hm, use_amp=True)
    m.net_AutoEncoder.bf16_storage = storage
    m.set_input(data)
    m.optimize_parameters()
    print("bf16_storage", storage, "loss", m.get_current_errors(), "ref", float(ref["loss_pose"]), float(ref["loss_cos_sim"]))
    for k, v in m.net_AutoEncoder.named_parameters():
        g = ref["grads"].get(k)
        if g is None or v.numel() < 65536:
            continue
        a, b = v.grad.double().reshape(-1).cpu(), g.double().reshape(-1)
        print(f"   {k[-60:]:60s} cos {float(a @ b / (a.norm() * b.norm())):.5f} rel {float((a - b).norm() / b.norm()):.3e} |g| {float(b.norm()):.3e}")
