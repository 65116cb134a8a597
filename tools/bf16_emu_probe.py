#!/usr/bin/env python3
"""How close is the HIP bf16-storage step to the float64 oracle WITH the bf16 rounding hook (oracle.lift_ref.Bf16Storage), and to the
plain float64 oracle?  Prints loss / pose / per-tensor gradient distances for a B = 2 step (UnrealEgo) -- the numbers the gates of
tests/test_gpu_configs.py are set from.  GPU box only.  usage: python tools/bf16_emu_probe.py [preset] [hm] [B]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from egotap_amd import spec                                     # noqa: E402
from egotap_amd.synthetic import synth_state_dict               # noqa: E402
from oracle import lift_ref as O                                # noqa: E402
from test_gpu_configs import _data, _model                      # noqa: E402


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "UnrealEgo"
    hm_size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    m, p = _model(preset, hm_size, use_amp=True)
    data, hm, gt = _data(B, p, "c3", gt_range=1.0)
    m.set_input(data)
    m.optimize_parameters()
    torch.cuda.synchronize()
    sd = O.to_torch_sd(synth_state_dict(spec.lift_state_spec(p)), torch.float64)
    refs = {"emulated": O.train_step(hm.double(), gt.double(), sd, p, round=O.Bf16Storage), "exact": O.train_step(hm.double(), gt.double(), sd, p)}
    errs = m.get_current_errors()
    pose = m.pred_pose.detach().double().cpu()
    for name, ref in refs.items():
        print(f"== vs {name} float64 oracle")
        print(f"  loss_pose rel {abs(errs['pose'] - float(ref['loss_pose'])) / abs(float(ref['loss_pose'])):.3e}   "
              f"loss_cos abs {abs(errs['cos_sim'] - float(ref['loss_cos_sim'])):.3e} (value {float(ref['loss_cos_sim']):.3e})")
        print(f"  pose max abs {float((pose - ref['pose']).abs().max()):.3e} (scale {float(ref['pose'].abs().max()):.3f})")
        rows = []
        gmax = max(float(g.norm()) for g in ref["grads"].values() if g is not None)
        for k, v in m.net_AutoEncoder.named_parameters():
            g = ref["grads"].get(k)
            if g is None or v.grad is None:
                continue
            a, b = v.grad.double().reshape(-1).cpu(), g.reshape(-1)
            rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
            cos = float(a @ b / (a.norm() * b.norm()).clamp_min(1e-30))
            rows.append((rel, cos, float(b.norm()) / gmax, v.numel(), k))
        rows.sort(reverse=True)
        for rel, cos, share, n, k in rows[:12]:
            print(f"  rel {rel:.3e} cos {cos:.6f} norm/max {share:.2e} numel {n:9d} {k}")
        big = [r for r in rows if r[3] >= 65536 and r[2] >= 1e-2]
        print(f"  large tensors: worst rel {max(r[0] for r in big):.3e}, worst cos {min(r[1] for r in big):.6f};  all tensors: median rel {sorted(r[0] for r in rows)[len(rows) // 2]:.3e}")


if __name__ == "__main__":
    main()
