#!/usr/bin/env python3
"""Headline benchmark: stereo frames/s of the heatmap -> 3D lift (BASELINE.json configs[1]:
UnrealEgo 16-joint lifting, batch 256 per GPU, fp32, forward only) on N MI355X of one node.

  python bench.py --gpus 1 --steps 10 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (EgoTAPAutoEncoder.forward through the C ABI / HIP kernels) over one
batch of 256 synthetic stereo heatmap sets already resident in HBM.  The batch shards by sample with no
data-path collective, so N GPUs run N shards: weak scaling; value = N*256*K / max-rank time.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 256 CUs @ 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (no sparsity)


def frac_fields(mode: str, algorithmic_tflops: float) -> dict:
    """roofline fraction of an end-to-end rate in precision mode `mode`, against the peak of the pipe the large products run on:
    f32 -> exact-fp32 MFMA; bf16 -> bf16 MFMA; bf16x3 -> bf16 MFMA with EXECUTED flops (three MFMAs per algorithmic product)."""
    if mode == "f32":
        return {"end_to_end_frac_of_f32_mfma_peak": round(algorithmic_tflops / PEAK_F32_MFMA_TFLOPS, 4)}
    ex = 3.0 if mode == "bf16x3" else 1.0
    return {"executed_bf16_tflops_per_gpu": round(ex * algorithmic_tflops, 2),
            "end_to_end_frac_of_bf16_mfma_peak": round(ex * algorithmic_tflops / PEAK_BF16_MFMA_TFLOPS, 4)}


def lift_flops_per_frame(p) -> float:
    """Algorithmic FLOPs (2 per MAC, matmul only) of one lifting-head forward: SURVEY.md 8(d)."""
    N, D, T, J, H = p.seq, p.vit_dim, p.tokens, p.n_joints_hm, p.pu_hidden
    patch = 2 * N * 256 * D
    vit = p.vit_layers * (24 * N * D * D + 4 * N * N * D)
    pos_fc = 2 * T * (p.ppd * p.ppd * D * 2048 + 2048 * 512 + 512 * p.hidden)
    rot_fc = 2 * T * (2 * p.hm_size ** 2 * 2048 + 2048 * 512 + 512 * p.hidden)
    x = 2 * p.hidden
    pu = 2 * J * (x * (H + x) + 2 * x * 4 * H + H * 4 * H + H * H + 2 * H * 4 * H)
    head = 2 * J * (x + H) * 3 + (2 * J * H * 6 if p.estimate_head else 0)
    return float(patch + vit + pos_fc + rot_fc + pu + head)


def hm_flops_per_frame(n_out: int, rgb: int = 256) -> float:
    """Algorithmic FLOPs of one heatmap estimator (both eyes), resnet18 U-Net of net_architecture.py:25-173."""
    f = 0.0
    s = rgb // 2
    f += 2 * 2.0 * 147 * 64 * s * s                                   # stem, two eyes
    cin, s = 64, rgb // 4
    for i, c in enumerate((64, 128, 256, 512)):
        if i > 0:
            s //= 2
        px = 2.0 * s * s                                               # two eyes
        f += 2 * 9 * cin * c * px + 3 * 2 * 9 * c * c * px             # 4 convs 3x3
        if i > 0:
            f += 2 * cin * c * px                                      # 1x1 downsample
        cin = c
    s8, s16, s32, s64 = rgb // 32, rgb // 16, rgb // 8, rgb // 4
    f += 2.0 * 1024 * 1024 * s8 * s8 + 2.0 * 512 * 516 * s16 * s16 + 2.0 * 256 * 256 * s32 * s32 + 2.0 * 128 * 128 * s64 * s64
    f += 2.0 * 9 * 1540 * 1024 * s16 * s16 + 2.0 * 9 * 1280 * 512 * s32 * s32 + 2.0 * 9 * 640 * 512 * s64 * s64
    f += 2.0 * 512 * n_out * s64 * s64
    return f


def host_cores() -> int:
    """CPU cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    env = os.environ.get("EGOTAP_CPU_THREADS")
    if env:
        n = int(env)
    return n


def cpu_baseline(p, sd_np, batch: int, reps: int):
    """The oracle (CPU restatement of the reference, plain torch ops) timed on the host cores."""
    import torch
    from egotap_amd.synthetic import synth_input
    from oracle import lift_ref as O
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = O.to_torch_sd(sd_np)
    hm = torch.from_numpy(synth_input("hm_cpu_baseline", (batch, p.in_channels, p.hm_size, p.hm_size)))
    with torch.no_grad():
        O.lift_forward(hm[:2], sd, p)      # warm-up
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = O.lift_forward(hm, sd, p)
            ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"value": batch / med, "unit": "stereo frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle/lift_ref.lift_forward, B={batch}, fp32, {reps} timed passes (median), torch CPU threads={cores}"}, out


def measured_traffic(by_kernel, batch):
    """HBM bytes per launch of the GEMM kernels from the PMC counters of the committed rocprofv3 run
    (profiles/rNN_traffic.json, written by tools/summarize_prof.py from separate --pmc FETCH_SIZE / WRITE_SIZE passes over
    this same command at B = 256), weighted by this run's launch counts.  None when no profile matches."""
    import glob
    import re
    if batch != 256:
        return None, None
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    prof = json.load(open(files[-1]))["kernels"]

    def key(name):
        base = name.split("<", 1)[0]
        ids = re.findall(r"(ALoad\w+|Epi\w+|bf16x3)", name)
        return (base,) + tuple(ids)
    table = {key(k): v for k, v in prof.items()}
    tot, n = 0.0, 0
    for name, v in by_kernel.items():
        t = table.get(key(name))
        if t is None:
            continue
        tot += v["launches"] * (t["read_bytes"] + t["write_bytes"])
        n += v["launches"]
    if n == 0:
        return None, None
    return tot / n, os.path.basename(files[-1])


def dist_backend():
    import torch.distributed as dist
    return dist.get_backend() if dist.is_initialized() else None


def timed_lift(net, hm, steps, warmup, lib, L, h, barrier, dev, timing, world):
    """W warm-up + K timed forwards bracketed by barrier + synchronize; returns (elapsed max over ranks, pose, timing detail)"""
    import torch
    from egotap_amd import parallel
    for _ in range(max(warmup, 1)):
        pose = net.predict_pose(hm)
    barrier()
    lib.check(L.egotap_timing_enable(h, int(timing)))
    t0 = time.perf_counter()
    for _ in range(steps):
        pose = net.predict_pose(hm)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    barrier()
    timed_lift.per_rank_ms = [round(1e3 * t / steps, 3) for t in parallel.gather_floats(elapsed, dev)]
    elapsed = parallel.max_over_ranks(elapsed, dev)
    out = None
    if timing:
        n, ms, fl = C.c_int(), C.c_double(), C.c_double()
        lib.check(L.egotap_timing_read(h, C.byref(n), C.byref(ms), C.byref(fl)))
        out = (n.value, ms.value, fl.value, json.loads(L.egotap_timing_detail(h).decode()))
        lib.check(L.egotap_timing_enable(h, 0))
    return elapsed, pose, out


def bench_config5(args, dev, rank, world, barrier, lib, L):
    """Secondary measurement: the geometry of BASELINE configs[4] -- EgoCap preset (17 heatmaps per eye, 17 joints) on 128x128
    heatmaps (512x512 RGB): 768^2 ViT image, 2304 tokens, fc1 K = 65536 / 32768 -- lifting head forward, fp32 and bf16x3."""
    import torch
    from egotap_amd import networks, spec
    from egotap_amd.options import preset_defaults
    from egotap_amd.synthetic import synth_input, synth_state_dict
    p5 = spec.lift_preset("EgoCap", 128)
    net = networks.EgoTAPAutoEncoder(preset_defaults("EgoCap", 128), input_channel_scale=2)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(p5)).items()})
    net = net.to(dev).eval()
    B = args.config5_batch
    hm = torch.from_numpy(synth_input(f"hm_c5_rank{rank}", (4, p5.in_channels, 128, 128))).to(dev).repeat(B // 4, 1, 1, 1).contiguous()
    h = net._ensure_handle()
    out = {"workload": f"EgoCap lifting head, heatmaps [B,102,128,128], batch {B} per GPU", "batch_per_gpu": B,
           "flops_per_frame": lift_flops_per_frame(p5)}
    ref = None
    for mode in ("f32", "bf16x3", "bf16"):       # bf16 = the arithmetic BASELINE configs[4] names (bf16 activations in HBM)
        net.set_precision(mode)
        el, pose, _ = timed_lift(net, hm, 3, 1, lib, L, h, barrier, dev, False, world)
        fps = world * B * 3 / el
        tf5 = fps * out["flops_per_frame"] / world / 1e12
        out[mode] = {"value": round(fps, 1), "unit": "stereo frames/s", "ms_per_step": round(1e3 * el / 3, 2),
                     "end_to_end_tflops_per_gpu": round(tf5, 2), **frac_fields(mode, tf5)}
        if ref is None:
            ref = pose
        else:
            out[mode]["max_abs_diff_vs_f32_mode"] = float((pose - ref).abs().max())
    del net, hm
    torch.cuda.empty_cache()
    # the whole path at this geometry: 512x512 RGB -> two EgoCap estimators (128x128 maps) -> head, bf16 arithmetic and fp32
    from egotap_amd import models
    from egotap_amd.synthetic import synth_hm_state_dict
    opt = preset_defaults("EgoCap", 128)
    opt.gpu_ids = [dev.index]
    m = models.create_model(opt)
    J = p5.n_joints_hm
    for name, sd in (("AutoEncoder", synth_state_dict(spec.lift_state_spec(p5))), ("HeatMap", synth_hm_state_dict(J, "hm_pos.")),
                     ("RotHeatMap", synth_hm_state_dict(2 * J, "hm_rot."))):
        getattr(m, "net_" + name).load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    Bf = min(B, 32)
    rgb = [torch.from_numpy(synth_input(f"rgb5_{sd_}_rank{rank}", (4, 3, 512, 512), -2.0, 2.0)).to(dev).repeat(Bf // 4, 1, 1, 1).contiguous()
           for sd_ in ("l", "r")]
    m.set_input({"input_rgb_left": rgb[0], "input_rgb_right": rgb[1]})
    m.eval()                                   # test.py -> utils/evaluate.py:93 (set_eval_mode() leaves net_RotHeatMap in its mode, as the reference does)
    flops_full = out["flops_per_frame"] + hm_flops_per_frame(2 * J, 512) + hm_flops_per_frame(4 * J, 512)
    out["full_pipeline_from_rgb_512"] = {"batch_per_gpu": Bf, "flops_per_frame": flops_full}
    for mode in ("f32", "bf16"):
        m.set_precision(mode)
        with torch.no_grad():
            m.forward(evaluate=True)
            barrier()
            t0 = time.perf_counter()
            for _ in range(2):
                m.forward(evaluate=True)
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
        barrier()
        from egotap_amd import parallel
        el = parallel.max_over_ranks(el, dev)
        fps = world * Bf * 2 / el
        tff = fps * flops_full / world / 1e12
        out["full_pipeline_from_rgb_512"][mode] = {"value": round(fps, 1), "unit": "stereo frames/s", "ms_per_step": round(1e3 * el / 2, 2),
                                                   "end_to_end_tflops_per_gpu": round(tff, 2), **frac_fields(mode, tff)}
    del m, rgb
    torch.cuda.empty_cache()
    return out


def bench_full(args, p, dev, rank, world, barrier, lib, L, mode="f32"):
    """Secondary measurement: the whole path from RGB (two heatmap estimators + lifting head), same batch."""
    import torch
    import torch.distributed as dist
    from egotap_amd import models, spec
    from egotap_amd.options import preset_defaults
    from egotap_amd.synthetic import synth_hm_state_dict, synth_input, synth_state_dict
    opt = preset_defaults(args.preset)
    opt.gpu_ids = [dev.index]
    m = models.create_model(opt)
    J = p.n_joints_hm
    for name, sd in (("AutoEncoder", synth_state_dict(spec.lift_state_spec(p))), ("HeatMap", synth_hm_state_dict(J, "hm_pos.")),
                     ("RotHeatMap", synth_hm_state_dict(2 * J, "hm_rot."))):
        getattr(m, "net_" + name).load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    B = args.batch
    S = 4 * p.hm_size
    # uniform RGB in the ImageNet-normalised range; one 8-frame block repeated (generation cost only)
    blk = min(B, 8)
    l = torch.from_numpy(synth_input(f"rgb_l_rank{rank}", (blk, 3, S, S), -2.0, 2.0)).to(dev).repeat((B + blk - 1) // blk, 1, 1, 1)[:B].contiguous()
    r = torch.from_numpy(synth_input(f"rgb_r_rank{rank}", (blk, 3, S, S), -2.0, 2.0)).to(dev).repeat((B + blk - 1) // blk, 1, 1, 1)[:B].contiguous()
    m.set_input({"input_rgb_left": l, "input_rgb_right": r})
    m.eval()                                   # test.py -> utils/evaluate.py:93 (set_eval_mode() leaves net_RotHeatMap in its mode, as the reference does)
    m.set_precision(mode)
    h = m.net_HeatMap._ensure_handle()
    with torch.no_grad():
        m.forward(evaluate=True)
        barrier()
        lib.check(L.egotap_timing_enable(h, 1))
        lib.check(L.egotap_timing_enable(m.net_RotHeatMap._ensure_handle(), 1))
        t0 = time.perf_counter()
        for _ in range(args.full_steps):
            m.forward(evaluate=True)
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
    barrier()
    from egotap_amd import parallel
    elapsed = parallel.max_over_ranks(elapsed, dev)
    roles = {}
    tot_ms = tot_fl = 0.0
    for net in (m.net_HeatMap, m.net_RotHeatMap):
        n, ms, fl = C.c_int(), C.c_double(), C.c_double()
        lib.check(L.egotap_timing_read(net._ensure_handle(), C.byref(n), C.byref(ms), C.byref(fl)))
        lib.check(L.egotap_timing_enable(net._ensure_handle(), 0))
        tot_ms += ms.value
        tot_fl += fl.value
        for d in json.loads(L.egotap_timing_detail(net._ensure_handle()).decode()):
            a = roles.setdefault(d["role"], {"ms": 0.0, "flops": 0.0, "launches": 0, "kernel": d["kernel"]})
            a["ms"] += d["ms"]; a["flops"] += d["flops"]; a["launches"] += d["launches"]
    flops_frame = lift_flops_per_frame(p) + hm_flops_per_frame(2 * J, S) + hm_flops_per_frame(4 * J, S)
    fps = world * B * args.full_steps / elapsed
    conv_tf = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    pose = m.pred_pose.detach().clone()
    del m, l, r
    torch.cuda.empty_cache()
    return {
        "_pose": pose, "dtype": mode,
        "value": round(fps, 1), "unit": "stereo frames/s", "ms_per_step": round(1e3 * elapsed / args.full_steps, 2),
        "steps": args.full_steps, "flops_per_frame": flops_frame,
        "end_to_end_tflops_per_gpu": round(fps * flops_frame / world / 1e12, 2),
        **frac_fields(mode, fps * flops_frame / world / 1e12),
        "conv_roofline": ({"bound": "mfma", "kernel": "conv kernels (all instantiations, algorithmic FLOPs)", "achieved": round(conv_tf, 2),
                           "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(conv_tf / PEAK_F32_MFMA_TFLOPS, 4),
                           "conv_share_of_step_time": round(tot_ms * 1e-3 / elapsed, 4)} if mode == "f32" else
                          {"bound": "mfma", "kernel": "conv kernels (all instantiations; the 3x3 stride-1 ones on conv_bf16_kernel as hi+lo "
                                                      "splits: EXECUTED = 3 x algorithmic on those, counted as 3 x for all -> an upper bound)",
                           "achieved_algorithmic": round(conv_tf, 2), "achieved": round(3 * conv_tf, 2), "peak": PEAK_BF16_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(3 * conv_tf / PEAK_BF16_MFMA_TFLOPS, 4),
                           "conv_share_of_step_time": round(tot_ms * 1e-3 / elapsed, 4)} if mode == "bf16x3" else
                          {"bound": "mfma", "kernel": "conv kernels on bf16 channels-last tensors (layers 2-4 and decoder: gemm_bf16s_kernel implicit GEMMs; "
                                                      "layer1: conv64_direct_bf16s_kernel; the fused stem + max-pool kernel is not in this sum), algorithmic FLOPs",
                           "achieved": round(conv_tf, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(conv_tf / PEAK_BF16_MFMA_TFLOPS, 4), "conv_share_of_step_time": round(tot_ms * 1e-3 / elapsed, 4)}),
        "by_role": {k: {"kernel": v["kernel"], "avg_ms": round(v["ms"] / v["launches"], 4),
                        "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in roles.items()},
    }


def bench_train(args, p, dev, rank, world, barrier, mode="f32", batch=None, from_rgb=False, reference_default_bn=False, preset=None, hm_size=None):
    """Secondary measurement: one optimisation step of the lifting head (train-mode forward from resident heatmaps,
    loss, backward, gradient all-reduce when world > 1, AdamW).  mode = arithmetic of the GEMMs (egotap_set_precision):
    f32 exact, bf16x3 split (fp32-grade gradients), bf16 (BASELINE config 3: bf16 MFMA, fp32 accumulate and master weights).
    from_rgb: the step as train.py runs it without --use_gt_heatmap -- RGB frames through the two frozen heatmap estimators
    (--path_to_trained_heatmap checkpoints, written here from the synthetic state_dicts; --use_amp arithmetic) and on into the head."""
    import tempfile
    import torch
    from egotap_amd import models, parallel, spec
    from egotap_amd.options import preset_defaults
    from egotap_amd.synthetic import synth_hm_state_dict, synth_input, synth_state_dict
    torch.cuda.empty_cache()
    held = torch.cuda.memory_allocated(dev)      # what the earlier legs of this process still hold (the headline net and its workspace)
    if preset is not None:                       # another preset / heatmap side than the headline's (config 5: EgoCap, 128 x 128 heatmaps)
        p = spec.lift_preset(preset, hm_size or 64)
    opt = preset_defaults(preset or args.preset, hm_size) if hm_size else preset_defaults(preset or args.preset)
    opt.gpu_ids, opt.isTrain, opt.use_gt_heatmap = [dev.index], True, not from_rgb
    opt.lr, opt.opt_eps, opt.weight_decay = 1e-3, 1e-4, 0.0
    tmp = None
    if from_rgb:
        tmp = tempfile.TemporaryDirectory()
        J0 = p.n_joints_hm
        for sub, n_hm, salt in (("hm_pos", J0, "hm_pos."), ("hm_" + getattr(opt, "heatmap_type", "sin"), 2 * J0, "hm_rot.")):
            os.makedirs(os.path.join(tmp.name, sub))
            torch.save({k: torch.from_numpy(v) for k, v in synth_hm_state_dict(n_hm, salt).items()}, os.path.join(tmp.name, sub, "best_net_HeatMap.pth"))
        opt.log_dir = tmp.name
        opt.path_to_trained_heatmap = os.path.join(tmp.name, "hm", "best_net_HeatMap.pth")
        opt.use_amp, opt.amp_precision = mode != "f32", (mode if mode != "f32" else "bf16")
        # the opt-out leg's frozen estimators use folded running-statistics BatchNorm (frames independent, the
        # bf16 channels-last kernels, chunks of 256 frames) and says so in its "input" field.  reference_default_bn: the wrapper's DEFAULT,
        # which is the reference's -- batch-statistics BatchNorm in the frozen estimators under model.train() (train.py:91), per eye, running
        # statistics drifting; under --use_amp on the bf16 channels-last kernels (egotap_hm_forward_bnbatch: backbone over the whole batch,
        # decoder in chunks of 256 frames).  [r5] measured BESIDE the opt-out leg.
        opt.frozen_heatmap_bn_eval = not reference_default_bn
    m = models.create_model(opt)
    m.net_AutoEncoder.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(p)).items()})
    m.net_AutoEncoder.set_precision(mode)
    torch.cuda.reset_peak_memory_stats(dev)
    B, J = batch or args.train_batch, p.n_joints_hm
    hm = torch.from_numpy(synth_input(f"hm_train_rank{rank}", (min(B, 16), p.in_channels, p.hm_size, p.hm_size))).to(dev)
    hm = hm.repeat((B + hm.shape[0] - 1) // hm.shape[0], 1, 1, 1)[:B].contiguous()
    gt = torch.from_numpy(synth_input(f"gt_train_rank{rank}", (B, p.out_joints, 3), -20.0, 20.0)).to(dev)
    data = {"input_rgb_left": torch.zeros(1, 3, 4, 4), "input_rgb_right": torch.zeros(1, 3, 4, 4), "gt_heatmap_left": hm[:, :J],
            "gt_heatmap_right": hm[:, J:2 * J], "gt_limb_heatmap_left": hm[:, 2 * J:4 * J], "gt_limb_heatmap_right": hm[:, 4 * J:],
            "gt_local_pose": gt}
    if from_rgb:
        S = 4 * p.hm_size
        for side in ("left", "right"):
            blk = torch.from_numpy(synth_input(f"rgb_{side}_train_rank{rank}", (8, 3, S, S), -2.0, 2.0)).to(dev)
            data["input_rgb_" + side] = blk.repeat((B + 7) // 8, 1, 1, 1)[:B].contiguous()
    m.set_input(data)
    if from_rgb and reference_default_bn:
        m.train()                                # train.py:91: the wrapper -- frozen estimators included -- in training mode
    for _ in range(2):
        m.optimize_parameters()                  # warm-up (allocator, BN buffers, optimizer state)
    barrier()
    ms0 = torch.cuda.memory_stats(dev)
    t0 = time.perf_counter()
    for _ in range(args.train_steps):
        m.optimize_parameters()
    torch.cuda.synchronize(dev)
    own = time.perf_counter() - t0
    per_rank = [round(1e3 * t / args.train_steps, 3) for t in parallel.gather_floats(own, dev)]
    elapsed = parallel.max_over_ranks(own, dev)
    ms1 = torch.cuda.memory_stats(dev)
    dev_allocs = ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)      # hipMalloc calls inside the timed region (should be 0)
    barrier()
    errs = m.get_current_errors()
    fps = world * B * args.train_steps / elapsed
    flops = 3.0 * lift_flops_per_frame(p)        # algorithmic: backward = 2 x forward (dgrad + wgrad)
    if from_rgb:                                 # + the frozen estimators' forward
        flops += hm_flops_per_frame(2 * p.n_joints_hm, 4 * p.hm_size) + hm_flops_per_frame(4 * p.n_joints_hm, 4 * p.hm_size)
    peak_gb = (torch.cuda.max_memory_allocated(dev) - held) / 2 ** 30
    red = m.net_AutoEncoder._reducer()
    exposed = red.read_exposed_ms() if world > 1 else None       # None at N = 1: there is no collective to be exposed
    ar_bytes, ar_buckets = (red.last_bytes, red.last_buckets) if world > 1 else (0, 0)
    del m, hm, gt, data
    from egotap_amd import training
    training.release_scratch()                   # (the model, with the activations' buffer it keeps, is gone already)
    torch.cuda.empty_cache()
    if tmp is not None:
        tmp.cleanup()
    return {"value": round(fps, 1), "unit": "stereo frames/s (training step)", "ms_per_step": round(1e3 * elapsed / args.train_steps, 2),
            "steps": args.train_steps, "batch_per_gpu": B, "dtype": mode, "flops_per_frame": flops,
            "end_to_end_tflops_per_gpu": round(fps * flops / world / 1e12, 2), **frac_fields(mode, fps * flops / world / 1e12),
            "loss_pose": errs.get("pose"), "loss_cos_sim": errs.get("cos_sim"), "peak_hbm_gib": round(peak_gb, 1),
            "allreduce_exposed_ms_last_step": None if exposed is None else round(exposed, 3),
            "allreduce_bytes_per_step": int(ar_bytes), "allreduce_buckets": int(ar_buckets),
            "allreduce_op": ("avg (RCCL ReduceOp.AVG, in place on the gradient arena)" if world > 1 and dist_backend() == "nccl" else
                             "sum + 1/world scaling pass (gloo)" if world > 1 else None),
            "per_rank_ms_per_step": per_rank, "device_allocations_in_timed_region": int(dev_allocs),
            "input": ("RGB frames through the two frozen heatmap estimators in TRAIN mode as train.py:91 leaves them (the reference's and the wrapper's "
                      "default: batch-statistics BatchNorm per eye, running statistics updated; --use_amp arithmetic on bf16 channels-last tensors), then the head"
                      if from_rgb and reference_default_bn else
                      "RGB frames through the two frozen heatmap estimators (opt.frozen_heatmap_bn_eval: running-statistics BatchNorm, "
                      "SURVEY Appendix D.5; --use_amp arithmetic), then the head" if from_rgb
                      else "resident heatmaps (--use_gt_heatmap): the frozen estimators are not run; the four ground-truth maps are channel slices of one tensor in "
                           "the head's layout, so the wrapper's concatenation is a view (round 5; a 1.4 GB torch.cat per 1024-frame step before: ~0.9 ms)"),
            "frozen_estimators_bn": ("batch statistics (reference default)" if reference_default_bn else "running statistics (opt-out)") if from_rgb else None,
            "preset": preset or args.preset, "hm_size": p.hm_size,
            "note": "gradient all-reduce (N > 1) overlapped with the backward, bucket by bucket, in place on a flat arena; the attention "
                    "backward recomputes the scores (two kernels, 7 MFMA products), not counted in flops_per_frame"}


def bench_stage1(args, p, dev, rank, world, barrier, mode="f32"):
    """Secondary measurement: one optimisation step of the stage-1 heatmap estimator (position net: train-mode forward with
    per-eye BatchNorm statistics, MSE losses, full backward through decoder + ResNet-18, Adam), fp32."""
    import torch
    from egotap_amd import models, parallel
    from egotap_amd.options import preset_defaults
    from egotap_amd.synthetic import synth_hm_state_dict, synth_input
    opt = preset_defaults(args.preset)
    opt.model, opt.isTrain, opt.gpu_ids, opt.num_rot_heatmap = "heatmap_shared", True, [dev.index], 0
    opt.lr, opt.weight_decay, opt.lambda_heatmap = 1e-3, 0.0, 1.0
    m = models.create_model(opt)
    J = p.n_joints_hm
    m.net_HeatMap.load_state_dict({k: torch.from_numpy(v) for k, v in synth_hm_state_dict(J, "hm_pos.").items()})
    m.net_HeatMap.set_precision(mode)
    B, S = args.stage1_batch, 4 * p.hm_size
    torch.cuda.reset_peak_memory_stats(dev)
    blk = min(B, 8)
    rep = (B + blk - 1) // blk
    data = {"input_rgb_left": torch.from_numpy(synth_input(f"s1_l_rank{rank}", (blk, 3, S, S), -2.0, 2.0)).to(dev).repeat(rep, 1, 1, 1)[:B],
            "input_rgb_right": torch.from_numpy(synth_input(f"s1_r_rank{rank}", (blk, 3, S, S), -2.0, 2.0)).to(dev).repeat(rep, 1, 1, 1)[:B],
            "gt_heatmap_left": torch.from_numpy(synth_input(f"s1_gl_rank{rank}", (blk, J, p.hm_size, p.hm_size))).to(dev).repeat(rep, 1, 1, 1)[:B],
            "gt_heatmap_right": torch.from_numpy(synth_input(f"s1_gr_rank{rank}", (blk, J, p.hm_size, p.hm_size))).to(dev).repeat(rep, 1, 1, 1)[:B]}
    m.set_input(data)
    m.optimize_parameters()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.train_steps):
        m.optimize_parameters()
    torch.cuda.synchronize(dev)
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    barrier()
    errs = m.get_current_errors()
    fps = world * B * args.train_steps / elapsed
    flops = 3.0 * hm_flops_per_frame(2 * J, S)
    peak_gb = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    del m, data
    torch.cuda.empty_cache()
    return {"value": round(fps, 1), "unit": "stereo frames/s (stage-1 training step, position net)", "ms_per_step": round(1e3 * elapsed / args.train_steps, 2),
            "steps": args.train_steps, "batch_per_gpu": B, "dtype": mode, "flops_per_frame": flops,
            "end_to_end_tflops_per_gpu": round(fps * flops / world / 1e12, 2), **frac_fields(mode, fps * flops / world / 1e12),
            "loss_heatmap_left": errs.get("heatmap_left"),
            "peak_hbm_gib": round(peak_gb, 1)}


def bench_latency_rgb(args, p, dev):
    """Serving-shaped leg: latency of the WHOLE path from RGB (two estimators + head, evaluate-style forward of the wrapper) at B = 1 and 8."""
    import torch
    from egotap_amd import models, spec
    from egotap_amd.options import preset_defaults
    from egotap_amd.synthetic import synth_hm_state_dict, synth_input, synth_state_dict
    opt = preset_defaults(args.preset)
    opt.gpu_ids = [dev.index]
    m = models.create_model(opt)
    J = p.n_joints_hm
    for name, sd in (("AutoEncoder", synth_state_dict(spec.lift_state_spec(p))), ("HeatMap", synth_hm_state_dict(J, "hm_pos.")),
                     ("RotHeatMap", synth_hm_state_dict(2 * J, "hm_rot."))):
        getattr(m, "net_" + name).load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.eval()
    S = 4 * p.hm_size
    out = {"unit": "ms per forward from RGB (both estimators + head; median of 12, synchronised)"}
    for B in (1, 8):
        l = torch.from_numpy(synth_input("rgb_l_lat", (B, 3, S, S), -2.0, 2.0)).to(dev)
        r = torch.from_numpy(synth_input("rgb_r_lat", (B, 3, S, S), -2.0, 2.0)).to(dev)
        m.set_input({"input_rgb_left": l, "input_rgb_right": r})
        for mode in ("f32", "bf16"):
            m.set_precision(mode)
            with torch.no_grad():
                for _ in range(3):
                    m.forward(evaluate=True)
                ts = []
                for _ in range(12):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    m.forward(evaluate=True)
                    torch.cuda.synchronize(dev)
                    ts.append(time.perf_counter() - t0)
            ts.sort()
            out[f"b{B}_{mode}"] = round(1e3 * ts[len(ts) // 2], 3)
    del m
    torch.cuda.empty_cache()
    return out


def bench_latency(net, p, dev):
    """Serving-shaped leg: latency of the lifting head at B = 1 and 8 (split-K small-batch GEMM path), per precision mode."""
    import torch
    from egotap_amd.synthetic import synth_input
    out = {"unit": "ms per forward (median of 20, synchronised)"}
    for B in (1, 8):
        hm = torch.from_numpy(synth_input("hm_lat", (B, p.in_channels, p.hm_size, p.hm_size))).to(dev)
        for mode in ("f32", "bf16x3", "bf16"):
            net.set_precision(mode)
            for _ in range(3):
                net.predict_pose(hm)
            ts = []
            for _ in range(20):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                net.predict_pose(hm)
                torch.cuda.synchronize(dev)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            out[f"b{B}_{mode}"] = round(1e3 * ts[len(ts) // 2], 3)
            if mode == "f32":                      # the same forward replayed from a captured HIP graph (one launch on the host side)
                for _ in range(3):
                    net.predict_pose_graphed(hm)
                ts = []
                for _ in range(20):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    net.predict_pose_graphed(hm)
                    torch.cuda.synchronize(dev)
                    ts.append(time.perf_counter() - t0)
                ts.sort()
                out[f"b{B}_{mode}_hip_graph"] = round(1e3 * ts[len(ts) // 2], 3)
    net.set_precision("f32")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="stereo frames per GPU per step (BASELINE: 256)")
    ap.add_argument("--preset", default="UnrealEgo")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=32, help="frames per timed CPU pass (3 passes + warm-up: ~15 s of CPU work)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket GEMM launches with HIP events")
    ap.add_argument("--lift-only", action="store_true", help="skip the secondary full-pipeline (RGB -> joints) measurement")
    ap.add_argument("--full-steps", type=int, default=3)
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the secondary bf16x3 fast-mode measurement")
    ap.add_argument("--train-steps", type=int, default=2, help="timed optimisation steps of the secondary training measurement (0 = skip)")
    ap.add_argument("--train-batch", type=int, default=256)
    ap.add_argument("--config5-batch", type=int, default=64, help="per-GPU batch of the EgoCap / 128x128-heatmap measurement (0 = skip)")
    ap.add_argument("--config5-train-batch", type=int, default=256, help="per-GPU batch of the EgoCap / 128x128-heatmap TRAINING step (BASELINE config 5)")
    ap.add_argument("--stage1-batch", type=int, default=32, help="per-GPU batch of the stage-1 heatmap-estimator training measurement")
    ap.add_argument("--train-batch-bf16", type=int, default=1024, help="per-GPU batch of the bf16 training measurement (BASELINE config 3)")
    ap.add_argument("--all-legs", action="store_true", help="N > 1: also run the single-GPU secondary legs on every rank (default at N > 1: "
                                                            "headline + the data-parallel training legs, the ones with a collective)")
    args = ap.parse_args()

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as CHILD processes, before anything in this
        # process has touched the GPU (never an exec from a process that has initialised HIP), and return the launcher's exit code.
        # The children check that N devices exist and fail with a clear message if not.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    import torch
    import torch.distributed as dist
    from egotap_amd import lib, spec
    from egotap_amd.synthetic import synth_input, synth_state_dict
    from egotap_amd.options import preset_defaults as make_opt
    from egotap_amd import networks

    rank = int(os.environ.get("RANK", "0"))
    world = world_env
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world                      # launched by torch.distributed.run: the launcher's world size is authoritative
    # rehearsal on a one-GPU box: EGOTAP_DIST_BACKEND=gloo lets several ranks share cuda:0 (RCCL needs one device per rank)
    backend = os.environ.get("EGOTAP_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()          # counting devices does not initialise HIP
    if n_dev == 0:
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    if backend == "nccl" and world > n_dev:
        raise SystemExit(f"bench.py --gpus {world} needs {world} devices (one rank per GPU over RCCL); this node has {n_dev} "
                         f"[rank {rank}] -- for a rehearsal with ranks sharing a GPU set EGOTAP_DIST_BACKEND=gloo")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    if backend != "nccl" and world > 1:
        os.environ["EGOTAP_SHARED_DEVICE"] = "1"      # ranks share one GPU: kernels that need the device to themselves are switched off
    if backend != "nccl":
        local = local % max(n_dev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from egotap_amd import parallel
    parallel.init_from_env(backend, dev)     # "nccl" is RCCL on ROCm; no-op for one rank
    if world > 1 and not args.all_legs:        # N > 1 measures what N > 1 changes: the sharded headline and the data-parallel training step
        args.lift_only_secondary = True
    else:
        args.lift_only_secondary = False

    p = spec.lift_preset(args.preset)
    sd_np = synth_state_dict(spec.lift_state_spec(p))
    net = networks.EgoTAPAutoEncoder(make_opt(args.preset), input_channel_scale=2)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    net = net.to(dev).eval()
    B = args.batch
    hm = torch.from_numpy(synth_input(f"hm_bench_rank{rank}", (B, p.in_channels, p.hm_size, p.hm_size))).to(dev)

    L = lib.load()
    h = net._ensure_handle()

    def barrier():
        parallel.barrier(dev)

    timing = not args.no_kernel_timing
    elapsed, pose, t1 = timed_lift(net, hm, args.steps, args.warmup, lib, L, h, barrier, dev, timing, world)
    headline_per_rank_ms = list(timed_lift.per_rank_ms)

    roof = None
    if timing:
        n_launch, ms_total, fl_total, detail = t1
        by_kernel = {}
        for d in detail:
            k = by_kernel.setdefault(d["kernel"], {"launches": 0, "ms": 0.0, "flops": 0.0})
            k["launches"] += d["launches"]; k["ms"] += d["ms"]; k["flops"] += d["flops"]
        achieved = fl_total / (ms_total * 1e-3) / 1e12 if ms_total > 0 else 0.0
        roof = {
            "bound": "mfma", "kernel": "fp32 GEMM kernels (gemm_f32_dma_kernel for the ViT projections + gemm_f32_persist_kernel / gemm_f32_kernel, v_mfma_f32_32x32x2_f32)",
            "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
            "algorithmic_bytes": None, "traffic_source": None,
            "launches": n_launch, "avg_launch_ms": round(ms_total / max(n_launch, 1), 4),
            "gemm_share_of_step_time": round(ms_total * 1e-3 / elapsed, 4),
            "by_kernel": {k: {"launches": v["launches"], "avg_ms": round(v["ms"] / v["launches"], 4),
                              "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in by_kernel.items()},
            "by_role": {d["role"]: {"avg_ms": round(d["ms"] / d["launches"], 4),
                                    "tflops": round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 2)} for d in detail},
        }

        traffic, src = measured_traffic(by_kernel, B)
        if traffic is not None:
            roof["traffic"] = round(traffic)
            roof["traffic_source"] = f"profiles/{src}: rocprofv3 --pmc FETCH_SIZE(x2)+WRITE_SIZE per launch, launch-weighted over the GEMM kernels"
        # compulsory fp32 bytes of the same launches: A (M*K, or the gathered heatmap bytes) + W (N*K) + C (M*N), launch-weighted
        M_, D_ = B * p.seq, p.vit_dim
        per_layer = [(M_, 3 * D_, D_), (M_, D_, D_), (M_, 4 * D_, D_), (M_, D_, 4 * D_)]
        shapes = [(M_, D_, 256)] + per_layer * p.vit_layers + [(B * p.tokens, 2048, p.ppd * p.ppd * D_), (B * p.tokens, 2048, 2 * p.hm_size ** 2)]
        roof["algorithmic_bytes"] = round(sum(4.0 * (m * k + n_ * k + m * n_) for m, n_, k in shapes) / len(shapes))

    # opt-in fast mode: same path with the large GEMMs on the bf16 matrix cores as hi+lo splits (egotap_set_precision)
    fast = None
    if not args.no_fast_mode:
        net.set_precision("bf16x3")
        try:
            el3, pose3, t3 = timed_lift(net, hm, args.steps, args.warmup, lib, L, h, barrier, dev, timing, world)
        finally:
            net.set_precision("f32")
        fps3 = world * B * args.steps / el3
        fast = {"value": round(fps3, 1), "unit": "stereo frames/s", "ms_per_step": round(1e3 * el3 / args.steps, 3),
                "dtype": "bf16x3 (large GEMMs and attention: fp32 operands split hi+lo in registers, 3 x v_mfma_f32_32x32x16_bf16 per product, "
                         "fp32 accumulate; fp32 tensors in HBM; LayerNorm, small GEMMs, PU chain, pose head stay exact fp32)",
                "max_abs_diff_vs_f32_mode": float((pose3 - pose).abs().max()),
                "speedup_vs_f32_mode": round(fps3 / (world * B * args.steps / elapsed), 3)}
        if t3 is not None and t3[1] > 0:
            fast["gemm_algorithmic_tflops"] = round(t3[2] / (t3[1] * 1e-3) / 1e12, 2)
            fast["gemm_executed_bf16_tflops"] = round(3 * t3[2] / (t3[1] * 1e-3) / 1e12, 2)
            fast["gemm_frac_of_bf16_mfma_peak"] = round(3 * t3[2] / (t3[1] * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)
            fast["gemm_share_of_step_time"] = round(t3[1] * 1e-3 / el3, 4)

    # reduced-precision inference: bf16 activations in HBM, LDS-DMA bf16 GEMMs, bf16 attention (EGOTAP_PREC_BF16; the wrapper's --use_amp
    # arithmetic).  Not an fp32 result (bf16 keeps 8 significant bits per operand): reported beside the headline, never in `value`.
    fast16 = None
    if not args.no_fast_mode:
        net.set_precision("bf16")
        try:
            el16, pose16, _ = timed_lift(net, hm, args.steps, args.warmup, lib, L, h, barrier, dev, False, world)
        finally:
            net.set_precision("f32")
        fps16 = world * B * args.steps / el16
        tf16 = fps16 * lift_flops_per_frame(p) / world / 1e12
        fast16 = {"value": round(fps16, 1), "unit": "stereo frames/s", "ms_per_step": round(1e3 * el16 / args.steps, 3),
                  "dtype": "bf16 (bf16 activations in HBM, bf16 MFMA with fp32 accumulation, fp32 residual stream / LayerNorm statistics / "
                           "small FC layers / PU chain / pose head)",
                  "end_to_end_tflops_per_gpu": round(tf16, 2), **frac_fields("bf16", tf16),
                  "max_abs_diff_vs_f32_mode": float((pose16 - pose).abs().max()),
                  "speedup_vs_f32_mode": round(fps16 / (world * B * args.steps / elapsed), 3)}

    def leg(fn, *a, **k):
        """secondary measurements never take the headline line down with them"""
        try:
            return fn(*a, **k)
        except Exception as exc:      # noqa: BLE001 - reported in the JSON line
            torch.cuda.empty_cache()
            return {"error": f"{type(exc).__name__}: {exc}"[:300]}

    single_gpu_legs = not args.lift_only and not args.lift_only_secondary
    full = None
    if single_gpu_legs:
        full = leg(bench_full, args, p, dev, rank, world, barrier, lib, L)
        if not args.no_fast_mode:
            ff = leg(bench_full, args, p, dev, rank, world, barrier, lib, L, mode="bf16x3")
            if "_pose" in ff and "_pose" in full:
                ff["max_abs_pose_diff_vs_f32_mode"] = float((ff["_pose"] - full["_pose"]).abs().max())
                ff["speedup_vs_f32_mode"] = round(ff["value"] / full["value"], 3)
            ff.pop("_pose", None)
            ff.pop("by_role", None)
            full["fast_mode_bf16x3"] = ff
            # reduced precision end to end: estimators with a bf16 channels-last decoder, head with bf16 activations (--use_amp arithmetic)
            fb = leg(bench_full, args, p, dev, rank, world, barrier, lib, L, mode="bf16")
            if "_pose" in fb and "_pose" in full:
                fb["max_abs_pose_diff_vs_f32_mode"] = float((fb["_pose"] - full["_pose"]).abs().max())
                fb["speedup_vs_f32_mode"] = round(fb["value"] / full["value"], 3)
            fb.pop("_pose", None)
            fb.pop("by_role", None)
            full["fast_mode_bf16"] = fb
        full.pop("_pose", None)

    config5 = None
    if single_gpu_legs and not args.no_fast_mode and args.config5_batch > 0:
        config5 = leg(bench_config5, args, dev, rank, world, barrier, lib, L)

    train = None
    if not args.lift_only and args.train_steps > 0:
        train = leg(bench_train, args, p, dev, rank, world, barrier)
        train["bf16x3"] = leg(bench_train, args, p, dev, rank, world, barrier, mode="bf16x3")
        # BASELINE configs[2] / [3]: UnrealEgo training step (fwd+bwd+AdamW), bf16, batch 1024 per GPU (x N GPUs, gradient all-reduce)
        train["config3_bf16_b1024"] = leg(bench_train, args, p, dev, rank, world, barrier, mode="bf16", batch=args.train_batch_bf16)
        if world > 1:
            train["config3_bf16_b1024"]["config4_global_batch"] = world * args.train_batch_bf16
        if single_gpu_legs:
            # the same step as train.py runs it without --use_gt_heatmap: RGB frames -> two frozen heatmap estimators -> head ("RGB for full")
            train["config3_bf16_b1024_from_rgb"] = leg(bench_train, args, p, dev, rank, world, barrier, mode="bf16", batch=args.train_batch_bf16,
                                                       from_rgb=True)
            # [r5] ... and as train.py REALLY runs it: the frozen estimators in train mode (batch-statistics BatchNorm), at config 3's batch and at the
            # reference's own batch (scripts/train/PoseEstimator/unrealego.sh: --batch_size 32), each beside its opt-out twin
            train["config3_bf16_b1024_from_rgb_reference_default"] = leg(bench_train, args, p, dev, rank, world, barrier, mode="bf16", batch=args.train_batch_bf16,
                                                                         from_rgb=True, reference_default_bn=True)
            a, b = train["config3_bf16_b1024_from_rgb_reference_default"], train["config3_bf16_b1024_from_rgb"]
            if "ms_per_step" in a and "ms_per_step" in b:
                a["ms_per_step_vs_opt_out"] = round(a["ms_per_step"] / b["ms_per_step"], 3)
            train["reference_batch_b32_bf16_from_rgb_reference_default"] = leg(bench_train, args, p, dev, rank, world, barrier, mode="bf16", batch=32, from_rgb=True,
                                                                               reference_default_bn=True)
            train["reference_batch_b32_bf16_from_rgb"] = leg(bench_train, args, p, dev, rank, world, barrier, mode="bf16", batch=32, from_rgb=True)
            # BASELINE config 5 "fwd (+ train step)": EgoCap, 128 x 128 heatmaps (512 x 512 RGB), bf16: the head's training step at the largest batch
            # one GPU holds comfortably (activations ~4 x config 3's per frame), and the same step from RGB with the estimators as train.py runs them
            if not args.no_fast_mode and args.config5_batch > 0:
                train["config5_egocap_hm128_bf16"] = leg(bench_train, args, p, dev, rank, world, barrier, mode="bf16", batch=args.config5_train_batch,
                                                          preset="EgoCap", hm_size=128)
                train["config5_egocap_hm128_bf16_from_rgb_reference_default"] = leg(bench_train, args, p, dev, rank, world, barrier, mode="bf16",
                                                                                    batch=min(64, args.config5_train_batch), from_rgb=True, reference_default_bn=True,
                                                                                    preset="EgoCap", hm_size=128)
            train["stage1_heatmap_estimator"] = leg(bench_stage1, args, p, dev, rank, world, barrier)
            if not args.no_fast_mode:
                train["stage1_heatmap_estimator"]["bf16x3"] = leg(bench_stage1, args, p, dev, rank, world, barrier, mode="bf16x3")
                # opt.amp_precision_heatmap = "bf16": one bf16 product per multiply on fp32 tensors (bf16-grade gradients: tests/test_gpu_hm_train_step.py)
                train["stage1_heatmap_estimator"]["bf16"] = leg(bench_stage1, args, p, dev, rank, world, barrier, mode="bf16")

    latency = None
    if rank == 0 and world == 1 and not args.lift_only:
        latency = leg(bench_latency, net, p, dev)
        if isinstance(latency, dict) and "error" not in latency:
            latency["from_rgb"] = leg(bench_latency_rgb, args, p, dev)

    cpu = None
    gpu_vs_oracle = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # rank 0 at N = 1 only
        cpu, ref = cpu_baseline(p, sd_np, args.cpu_batch, 3)
        # parity spot check on the same inputs the CPU leg used (first frames)
        hm_c = torch.from_numpy(synth_input("hm_cpu_baseline", (args.cpu_batch, p.in_channels, p.hm_size, p.hm_size))).to(dev)
        got = net.predict_pose(hm_c).cpu()
        gpu_vs_oracle = float((got - ref).abs().max())
        if fast is not None:
            net.set_precision("bf16x3")
            fast["max_abs_diff_vs_oracle"] = float((net.predict_pose(hm_c).cpu() - ref).abs().max())
            net.set_precision("f32")
        if fast16 is not None:
            net.set_precision("bf16")
            fast16["max_abs_diff_vs_oracle"] = float((net.predict_pose(hm_c).cpu() - ref).abs().max())
            fast16["oracle_pose_scale"] = float(ref.abs().max())
            net.set_precision("f32")

    parity_note = None
    if rank == 0 and world > 1:
        # N > 1: no timed CPU baseline (it is a property of the host, measured at N = 1), but the sharded headline still gets its numerical
        # check: rank 0's shard, first frames, against the oracle on the same inputs (a few seconds of CPU work)
        from oracle import lift_ref as O
        n_chk = min(4, B)
        with torch.no_grad():
            ref = O.lift_forward(hm[:n_chk].cpu(), O.to_torch_sd(sd_np), p)
        gpu_vs_oracle = float((net.predict_pose(hm[:n_chk]).cpu() - ref).abs().max())
        parity_note = f"rank 0's shard, first {n_chk} frames of the timed input, oracle/lift_ref.lift_forward on the host (cpu_baseline itself is timed at N = 1 only)"
    elif rank == 0 and gpu_vs_oracle is not None:
        parity_note = f"the cpu_baseline sample (B = {args.cpu_batch})"
    if rank == 0:
        frames = world * B * args.steps
        fps = frames / elapsed
        flops_frame = lift_flops_per_frame(p)
        line = {
            "metric": "stereo frames/sec (2D->3D lift, B=256, 256x256)", "value": round(fps, 1), "unit": "stereo frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.preset} 16-joint lifting head (heatmaps [B,90,64,64] -> joints [B,16,3]), "
                                   f"forward only, fp32, batch {B} per GPU, inputs resident in HBM",
                       "batch_per_gpu": B, "global_batch": world * B, "heatmap": p.hm_size, "rgb": 4 * p.hm_size,
                       "parallelism": f"dp{world} (batch shards, no collective)"},
            "flops_per_frame": flops_frame,
            "end_to_end_tflops_per_gpu": round(fps * flops_frame / world / 1e12, 2),
            "end_to_end_frac_of_f32_mfma_peak": round(fps * flops_frame / world / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "distributed": {"backend": dist_backend(), "world_size_seen": dist.get_world_size() if dist.is_initialized() else 1,
                            "launched_as": "torch.distributed.run, one rank per GPU" if world > 1 else "single process",
                            "per_rank_ms_per_step": headline_per_rank_ms,
                            "data_path_collective": "none (batch shards; samples are independent in eval)",
                            "legs_at_this_n": "all" if single_gpu_legs or args.lift_only else
                                              "headline + fast modes + data-parallel training legs (single-GPU legs: run with --gpus 1 or --all-legs)"},
            "roofline": roof, "cpu_baseline": cpu, "max_abs_diff_vs_oracle": gpu_vs_oracle,
            "parity_checked_on": parity_note if parity_note is not None else "not checked in this run (--no-cpu-baseline at N = 1)",
            "fast_mode_bf16x3": fast, "fast_mode_bf16": fast16, "full_pipeline_from_rgb": full, "config5_geometry_egocap_hm128": config5,
            "train_step_lifting_head": train, "small_batch_latency": latency,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
