#!/usr/bin/env python3
"""GPU probe: lifting-head training step in a precision mode at batch B (bench.py's bench_train leg alone).
usage: python tools/train_bf16_probe.py [B] [mode[,mode...]] [repeats]"""
import sys, os, json, argparse
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from egotap_amd import spec
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
modes = sys.argv[2] if len(sys.argv) > 2 else "bf16"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
args = argparse.Namespace(preset="UnrealEgo", train_steps=2, train_batch=B)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for mode in modes.split(","):          # several modes: one after the other in this process, as bench.py's legs run
    for _ in range(reps):
        r = bench.bench_train(args, spec.lift_preset("UnrealEgo"), dev, 0, 1, lambda: torch.cuda.synchronize(dev), mode=mode, batch=B)
        print(json.dumps({"lib": os.environ.get("EGOTAP_LIB", "shipped"), "B": B, "mode": mode, "frames_per_s": r["value"], "ms_per_step": r["ms_per_step"],
                          "loss_pose": r["loss_pose"], "peak_hbm_gib": r["peak_hbm_gib"], "tflops": r["end_to_end_tflops_per_gpu"]}), flush=True)
