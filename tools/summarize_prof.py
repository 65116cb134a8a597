#!/usr/bin/env python3
"""Digest gpurun_out/prof (written by tools/profile_bench.sh on the GPU box) into profiles/<tag>_*.

usage: python tools/summarize_prof.py r01
Copies the rocprofv3 kernel stats CSV and writes a markdown summary with per-kernel average
duration, HBM bytes per launch (FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md, WRITE_SIZE as is;
both in KiB), effective clock (GRBM_GUI_ACTIVE / 8 / duration) and matrix-pipe utilisation
(SQ_VALU_MFMA_BUSY_CYCLES / (clock cycles x 1024 SIMDs))."""
import collections
import csv
import glob
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gpurun_out", "prof")
HEADLINE_CMD = "python3 bench.py --steps 5 --warmup 2 --lift-only --no-fast-mode --no-cpu-baseline --no-kernel-timing"
CONFIG3_CMD = "python3 tools/train_bf16_probe.py 1024 bf16"
ALL_CMD = "python3 bench.py --steps 2 --warmup 1 --full-steps 1 --train-steps 1 --no-cpu-baseline --no-kernel-timing"


def newest(pat):
    """gpurun merges every call's files into gpurun_out/: keep only the newest file of a pass"""
    fs = sorted(glob.glob(os.path.join(SRC, pat)), key=os.path.getmtime)
    return fs[-1:] if fs else []


def rows(pat):
    out = []
    for f in newest(pat):
        out += list(csv.DictReader(open(f)))
    return out


def short(name):
    name = name.replace("void ", "")
    cut = name.find(">(")
    if cut > 0:
        name = name[:cut + 1]
    cut = name.find("(")
    if cut > 0 and "<" not in name[:cut]:
        name = name[:cut]
    return name.replace("GemmCfg", "").replace("PipeCfg", "Pipe")


def main():
    """profiles/<tag>_*: the HEADLINE command (what bench.py times: fp32 lifting head at B = 256 alone; per-launch HBM bytes are
    meaningful because every launch of a kernel has the same shape); profiles/<tag>_config3_*: one bf16 training step at B = 1024
    (BASELINE config 3) alone; profiles/<tag>_all_legs_*: every secondary leg as well (launches of one kernel mix batch sizes there)."""
    global SRC
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    if os.path.isdir(SRC):
        one(tag, HEADLINE_CMD, "B = 256; the timed workload of bench.py alone: fp32 lifting head forward")
    SRC = os.path.join(REPO, "gpurun_out", "prof_c3")
    if os.path.isdir(SRC):
        one(tag + "_config3", CONFIG3_CMD, "BASELINE config 3: UnrealEgo training step (forward + backward + AdamW), bf16 storage, batch 1024, 1 warm-up + 2 "
            "timed steps; per-launch HBM bytes are meaningful (one batch size)")
    SRC = os.path.join(REPO, "gpurun_out", "prof_hm")
    if os.path.isdir(SRC):
        one(tag + "_estimators_bf16", "python3 tools/hm_bf16_probe.py 256 64 bf16", "both heatmap estimators, 256 stereo frames of 256x256 RGB, EGOTAP_PREC_BF16: "
            "fused stem + max-pool on the bf16 matrix cores (stem_bf16s.h) -> layer1 on the direct halo-tile kernel (conv64_bf16s.h) -> layers 2-4 and the "
            "U-Net decoder as implicit GEMMs on bf16 channels-last tensors (conv_bf16s.h); "
            "2 warm-up + 3 timed passes x 2 nets", traffic_json=False)
    SRC = os.path.join(REPO, "gpurun_out", "prof_st1")
    if os.path.isdir(SRC):
        one(tag + "_stage1_f32", "python3 tools/stage1_probe.py 32 f32", "stage-1 training step of the position heatmap estimator, fp32, B = 32 stereo frames: "
            "train-mode forward, loss, backward, Adam; 1 warm-up + 2 timed steps", traffic_json=False)
    SRC = os.path.join(REPO, "gpurun_out", "prof_rgbdef")
    if os.path.isdir(SRC):
        one(tag + "_from_rgb_default", "python3 tools/from_rgb_probe.py 1024 default", "[r5] the lifting head's training step FROM RGB as train.py runs it: the two "
            "frozen estimators in train mode (batch-statistics BatchNorm per eye on the bf16 channels-last kernels: stem statistics pass + fused stem, "
            "bn_colstats / bn_finish / bn_apply per BatchNorm, backbone over the whole batch, decoder in chunks of 256), then the bf16-storage step of the head; "
            "B = 1024, 2 warm-up + 2 timed steps", traffic_json=False)
    SRC = os.path.join(REPO, "gpurun_out", "prof_all")
    if os.path.isdir(SRC):
        one(tag + "_all_legs", ALL_CMD, "every leg: fp32 headline, bf16x3 / bf16 fast modes, full pipeline in both modes, EgoCap / 128x128 geometry, "
            "training steps in f32 / bf16x3 / bf16 (B = 1024), stage-1 training, small-batch latency -- launches of one kernel mix batch sizes, "
            "so the per-launch averages here are not comparable with the headline table", traffic_json=False)


def one(tag, cmd, what, traffic_json=True):
    dst = os.path.join(REPO, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = rows("trace/*/*_kernel_stats.csv")
    for f in newest("trace/*/*_kernel_stats.csv"):
        shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    pm = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for x in rows("pmc_mfma/*/*_counter_collection.csv"):
        k = short(x["Kernel_Name"])
        pm[k][x["Counter_Name"]] += float(x["Counter_Value"])
        if x["Counter_Name"] == "GRBM_GUI_ACTIVE":
            pm[k]["ns"] += int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
            cnt[k] += 1
    fetch, write = collections.defaultdict(list), collections.defaultdict(list)
    for x in rows("pmc_fetch/*/*_counter_collection.csv"):
        fetch[short(x["Kernel_Name"])].append(float(x["Counter_Value"]))
    for x in rows("pmc_write/*/*_counter_collection.csv"):
        write[short(x["Kernel_Name"])].append(float(x["Counter_Value"]))
    lines = [f"# rocprofv3 summary {tag}", "",
             f"Command: `{cmd}` ({what}) under",
             "`rocprofv3 --kernel-trace --stats` (durations) and separate `--pmc` passes (FETCH_SIZE; WRITE_SIZE; "
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY).",
             "HBM read MB = FETCH_SIZE[KiB] x 2 / 1024 (gfx950 counts 64 B per 128-B request), write MB = WRITE_SIZE[KiB] / 1024.", "",
             "| kernel | calls | avg ms | % time | HBM read MB/launch | HBM write MB/launch | clock GHz | MFMA busy |",
             "|---|---|---|---|---|---|---|---|"]
    for s in stats:
        k = short(s["Name"])
        avg = float(s["AverageNs"]) / 1e6
        fr = 2 * sum(fetch[k]) / max(len(fetch[k]), 1) / 1024 if fetch[k] else float("nan")
        wr = sum(write[k]) / max(len(write[k]), 1) / 1024 if write[k] else float("nan")
        v = pm.get(k)
        clk = v["GRBM_GUI_ACTIVE"] / 8 / v["ns"] if v and v["ns"] else float("nan")
        busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024) if v and v["GRBM_GUI_ACTIVE"] else float("nan")
        lines.append(f"| `{k}` | {s['Calls']} | {avg:.4f} | {float(s['Percentage']):.2f} | {fr:.1f} | {wr:.1f} | {clk:.2f} | {busy:.3f} |")
    import json
    traffic = {}
    for s in stats:
        k = short(s["Name"])
        if fetch[k] and write[k]:
            traffic[k] = {"read_bytes": 2 * sum(fetch[k]) / len(fetch[k]) * 1024, "write_bytes": sum(write[k]) / len(write[k]) * 1024,
                          "launches_profiled": len(fetch[k]), "avg_ms": float(s["AverageNs"]) / 1e6}
    if traffic_json:
      json.dump({"tag": tag, "note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE (x2, gfx950) / WRITE_SIZE, separate passes; "
               "B = 256 per GPU", "kernels": traffic}, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
    open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main()
