#!/usr/bin/env python3
"""GPU probe: the lifting head's training step FROM RGB (bench.py's bench_train leg alone) with the frozen estimators either as train.py runs
them (train mode: batch-statistics BatchNorm, the wrapper's default) or on running statistics (opt.frozen_heatmap_bn_eval).
usage: python tools/from_rgb_probe.py [B] [default|optout|both] [mode]"""
import sys, os, json, argparse
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from egotap_amd import spec
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
which = sys.argv[2] if len(sys.argv) > 2 else "both"
mode = sys.argv[3] if len(sys.argv) > 3 else "bf16"
args = argparse.Namespace(preset="UnrealEgo", train_steps=2, train_batch=B)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for ref_default in ([True] if which == "default" else [False] if which == "optout" else [False, True]):
    r = bench.bench_train(args, spec.lift_preset("UnrealEgo"), dev, 0, 1, lambda: torch.cuda.synchronize(dev), mode=mode, batch=B, from_rgb=True,
                          reference_default_bn=ref_default)
    print(json.dumps({"B": B, "mode": mode, "frozen_estimators_bn": r["frozen_estimators_bn"], "frames_per_s": r["value"], "ms_per_step": r["ms_per_step"],
                      "peak_hbm_gib": r["peak_hbm_gib"]}), flush=True)
