#!/usr/bin/env python3
"""GPU probe: lifting-head training step in a precision mode at batch B (bench.py's bench_train leg alone)."""
import sys, os, json, argparse
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from egotap_amd import spec
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
args = argparse.Namespace(preset="UnrealEgo", train_steps=2, train_batch=B)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
gens = [int(g) for g in sys.argv[3].split(",")] if len(sys.argv) > 3 else [None]      # e.g. "1,2,1,2": attention kernel generations, interleaved
from egotap_amd import lib as _lib
for mode in mode.split(","):          # several modes: one after the other in this process, as bench.py's legs run
  for gen in gens:
    if gen is not None:
        _lib.check(_lib.load().egotap_debug_attention_gen(gen))
    r = bench.bench_train(args, spec.lift_preset("UnrealEgo"), dev, 0, 1, lambda: torch.cuda.synchronize(dev), mode=mode, batch=B)
    print(json.dumps({"B": B, "mode": mode, "attention_gen": gen, "frames_per_s": r["value"], "ms_per_step": r["ms_per_step"],
                      "loss_pose": r["loss_pose"], "peak_hbm_gib": r["peak_hbm_gib"], "tflops": r["end_to_end_tflops_per_gpu"]}))
