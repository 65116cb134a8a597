"""Race screen at the bench's sizes (development probe, GPU): everything below must be bit-identical from run to run.
  * the bf16-storage training step at B = 1024 (config 3): two fresh models, one step each -> losses and every gradient;
    then three more steps on the first model against the same three on the second
  * bf16 inference at B = 256, fp32 inference at B = 1 / 2 / 8 / 20 (the serving-batch routing of round 4) and the bf16 estimators at
    B = 136 / 256 (layer2-4 on the 64-deep GEMM): five or six forwards each
The hand-counted vmcnt waits of the DMA-staged kernels are the reason this exists: a wait that is one too generous shows up as a rare
run-to-run difference at full occupancy, not in the small parity cases."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
from egotap_amd.synthetic import synth_input  # noqa: E402
from gpu_util import hm_net, lift_net  # noqa: E402
from test_gpu_configs import _data, _model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
bad = 0


def grads(m):
    return {k: v.grad.detach().clone() for k, v in m.net_AutoEncoder.named_parameters() if v.grad is not None}


runs = []
for rep in range(2):
    m, p = _model(use_amp=True)
    data, hm, gt = _data(B, p, "det")
    m.set_input(data)
    rec = []
    for it in range(4):
        m.optimize_parameters()
        torch.cuda.synchronize()
        rec.append((dict(m.get_current_errors()), grads(m)))
    runs.append(rec)
    del m
    torch.cuda.empty_cache()
for it in range(4):
    (e0, g0), (e1, g1) = runs[0][it], runs[1][it]
    diff = [k for k in g0 if not torch.equal(g0[k], g1[k])]
    same_loss = all(e0[k] == e1[k] for k in e0)
    print(f"config 3 (B = {B}) step {it}: {len(g0)} gradient tensors, {len(diff)} differ, losses equal: {same_loss}", diff[:4])
    bad += len(diff) + (0 if same_loss else 1)
del runs
torch.cuda.empty_cache()

net, sd, p = lift_net("UnrealEgo")
hmx = torch.from_numpy(synth_input("hm_det", (8, p.in_channels, 64, 64))).cuda().repeat(32, 1, 1, 1).contiguous()
net.set_precision("bf16")
outs = [net.predict_pose(hmx).clone() for _ in range(5)]
nd = sum(0 if torch.equal(outs[0], o) else 1 for o in outs[1:])
print(f"bf16 inference B = 256: {nd} of 4 repeats differ")
bad += nd
net.set_precision("f32")
for Bs in (1, 2, 8, 20):        # serving batches in fp32: split-K GEMMs, the 64-row tile, key-split attention, the fused reduce + LayerNorm
    hs = torch.from_numpy(synth_input("hm_det_small", (Bs, p.in_channels, 64, 64))).cuda()
    outs = [net.predict_pose(hs).clone() for _ in range(6)]
    nd = sum(0 if torch.equal(outs[0], o) else 1 for o in outs[1:])
    print(f"fp32 inference B = {Bs}: {nd} of 5 repeats differ")
    bad += nd
for Bh in (136, 256):
    est, _ = hm_net("rot")
    l = torch.from_numpy(synth_input("det_l", (8, 3, 256, 256), -2.0, 2.0)).cuda().repeat(Bh // 8, 1, 1, 1).contiguous()
    r = torch.from_numpy(synth_input("det_r", (8, 3, 256, 256), -2.0, 2.0)).cuda().repeat(Bh // 8, 1, 1, 1).contiguous()
    est.set_precision("bf16")
    outs = [est(l, r).clone() for _ in range(5)]
    nd = sum(0 if torch.equal(outs[0], o) else 1 for o in outs[1:])
    same_frames = torch.equal(outs[0][0], outs[0][8])
    print(f"bf16 estimator B = {Bh}: {nd} of 4 repeats differ; frames 0 and 8 (the same image pair) equal: {same_frames}")
    bad += nd + (0 if same_frames else 1)
    del est, outs
    torch.cuda.empty_cache()
# [r5] the batch-statistics forward of a frozen estimator (egotap_hm_forward_bnbatch): the stem's statistics pass, the column-statistics /
# finish / apply kernels, the backbone over the whole batch and the chunked decoder -- from the same buffers the same bits, five times,
# at the bench's batch (1024 frames, chunks of 256) and at a batch that leaves the last chunk ragged
for Bh, chunk in ((1024, 256), (200, 64)):
    est, _ = hm_net("pos")
    est.set_precision("bf16")
    l = torch.from_numpy(synth_input("det_bn_l", (8, 3, 256, 256), -2.0, 2.0)).cuda().repeat(Bh // 8, 1, 1, 1).contiguous()
    r = torch.from_numpy(synth_input("det_bn_r", (8, 3, 256, 256), -2.0, 2.0)).cuda().repeat(Bh // 8, 1, 1, 1).contiguous()
    bufs0 = {k: v.detach().clone() for k, v in est.named_buffers()}
    outs, stats = [], []
    for _ in range(5):
        with torch.no_grad():
            for k, v in est.named_buffers():
                v.copy_(bufs0[k])
        o = torch.empty((Bh, 2 * est.num_heatmap, 64, 64), device="cuda")
        est.forward_bnbatch_into(l, r, o, chunk=chunk)
        torch.cuda.synchronize()
        outs.append(o)
        stats.append({k: v.detach().clone() for k, v in est.named_buffers()})
    nd = sum(0 if torch.equal(outs[0], o) else 1 for o in outs[1:])
    ns = sum(0 if all(torch.equal(stats[0][k], st[k]) for k in st) else 1 for st in stats[1:])
    print(f"batch-statistics bf16 estimator B = {Bh}, chunk {chunk}: {nd} of 4 repeats differ in the heatmaps, {ns} in the running statistics; finite: {bool(torch.isfinite(outs[0]).all())}")
    bad += nd + ns + (0 if bool(torch.isfinite(outs[0]).all()) else 1)
    with torch.no_grad():
        for k, v in est.named_buffers():
            v.copy_(bufs0[k])
    est.set_precision("f32")
    del est, outs, stats, l, r
    torch.cuda.empty_cache()
print("DETERMINISM", "OK" if bad == 0 else f"FAILED ({bad})")
sys.exit(0 if bad == 0 else 1)
