"""GPU parity of the lifting-head forward (C ABI, HIP kernels) against the golden vectors captured
from the reference and against the float64 oracle."""
import os

import numpy as np
import pytest
import torch

from egotap_amd.synthetic import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4     # north star: 3D joints within 1e-4 of the reference (fp32)


def _sample(t, stride=997):
    return t.reshape(-1)[::stride].cpu().numpy()


@pytest.mark.parametrize("tag,preset", [("ue", "UnrealEgo"), ("ec", "EgoCap")])
def test_lift_forward_matches_reference_golden(tag, preset):
    from gpu_util import lift_net
    g = np.load(os.path.join(GOLD, f"lift_fwd_{tag}_b2.npz"))
    net, _, p = lift_net(preset)
    hm = torch.from_numpy(synth_input(f"hm_{tag}", (2, p.in_channels, 64, 64))).cuda()
    pose, rot, indep, out_hm = net(hm)
    torch.cuda.synchronize()
    assert tuple(pose.shape) == (2, p.out_joints, 3)
    np.testing.assert_allclose(pose.cpu().numpy(), g["pose"], atol=TOL, rtol=0)
    J, T = p.n_joints_hm, p.tokens
    np.testing.assert_allclose(net.intermediate("pos_embed", 2).cpu().numpy().reshape(2, -1), g["pos_embed"], atol=TOL)
    np.testing.assert_allclose(net.intermediate("rot_embed", 2).cpu().numpy().reshape(2, -1), g["rot_embed"], atol=TOL)
    np.testing.assert_allclose(net.intermediate("skel_embed", 2).cpu().numpy().reshape(J, 2, 512), g["skel_embed"], atol=TOL)
    np.testing.assert_allclose(_sample(net.intermediate("tokens", 2)), g["final_ln_sample"], atol=TOL)
    # the reference's all-zero outputs keep their shapes
    assert tuple(rot.shape) == (2, 3 * J) and tuple(indep.shape) == (2, 6 * J) and tuple(out_hm.shape) == tuple(hm.shape)
    assert float(rot.abs().max()) == 0 and float(indep.abs().max()) == 0 and float(out_hm.abs().max()) == 0
    # tighter: the error should be fp32 rounding, orders below the gate
    assert np.abs(pose.cpu().numpy() - g["pose"]).max() < 2e-5


def test_vit_hidden_states_match_golden():
    import ctypes as C
    from egotap_amd import lib
    from gpu_util import lift_net
    g = np.load(os.path.join(GOLD, "lift_fwd_ue_b2.npz"))
    net, _, p = lift_net("UnrealEgo")
    hm = torch.from_numpy(synth_input("hm_ue", (2, p.in_channels, 64, 64))).cuda()
    L = lib.load()
    try:
        for stage, name in ((1, "emb"), (2, "layer0"), (3, "layer1"), (4, "layer2")):
            lib.check(L.egotap_lift_debug_stop(net._ensure_handle(), stage))
            net(hm)
            torch.cuda.synchronize()
            x = net.intermediate("x", 2)
            np.testing.assert_allclose(_sample(x), g[name + "_sample"], atol=TOL, err_msg=name)
            s = x.double()
            np.testing.assert_allclose([s.sum().item(), s.abs().sum().item()], g[name + "_stats"], rtol=1e-5)
    finally:
        lib.check(L.egotap_lift_debug_stop(net._ensure_handle(), 0))


@pytest.mark.parametrize("preset,B", [("UnrealEgo", 3), ("EgoCap", 1), ("UnrealEgo", 5), ("UnrealEgo", 1), ("UnrealEgo", 2)])      # (B = 1 / 2: key-split attention, 6 / 3 ranges)
def test_lift_forward_matches_oracle(preset, B):
    from gpu_util import lift_net
    from oracle import lift_ref as O
    net, sd_np, p = lift_net(preset)
    hm = torch.from_numpy(synth_input(f"hm_oracle_{preset}_{B}", (B, p.in_channels, 64, 64)))
    sd = O.to_torch_sd(sd_np, torch.float64)
    with torch.no_grad():
        ref = O.lift_forward(hm.double(), sd, p)
    pose = net.predict_pose(hm.cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(pose.cpu().numpy(), ref.numpy(), atol=TOL, rtol=0)


@pytest.mark.parametrize("preset,hm,B", [("UnrealEgo", 32, 3), ("UnrealEgo", 96, 2), ("EgoCap", 48, 2), ("UnrealEgo", 16, 4), ("UnrealEgo", 96, 40)])
def test_lift_forward_at_heatmap_sides_whose_sequence_is_not_a_multiple_of_32(preset, hm, B):
    """The reference accepts every heatmap side that is a multiple of 16 (net_architecture.py:327); the ViT sequence is then 36 (hm / 16)^2 tokens:
    36, 144, 324, 1296 at 16 / 32 / 48 / 96 -- no multiple of 32.  The exact-fp32 attention kernel masks the ragged last key tile and overlaps
    the last query block (attention_f32.h); everything else of the head is size-generic.  fp32 against the float64 oracle at the north-star
    tolerance; the bf16 modes take the fp32 attention kernel BY NAME for such sequences (egotap.h egotap_attention) and are bf16-grade.
    B = 40 at 96 x 96: batches past the serving split paths (51840 token rows)."""
    from gpu_util import lift_net
    from oracle import lift_ref as O
    net, sd_np, p = lift_net(preset, hm)
    assert p.seq % 32 != 0
    nb = min(B, 4)
    hmap = torch.from_numpy(synth_input(f"hm_side_{preset}_{hm}", (nb, p.in_channels, hm, hm)))
    if B > nb:
        hmap = (hmap[torch.arange(B) % nb] * (1.0 + 0.05 * (torch.arange(B) // nb).float()).view(B, 1, 1, 1)).contiguous()
    sd = O.to_torch_sd(sd_np, torch.float64)
    check = list(range(min(B, 3))) + ([B - 1] if B > 3 else [])
    with torch.no_grad():
        ref = O.lift_forward(hmap[check].double(), sd, p)
    pose = net.predict_pose(hmap.cuda())
    again = net.predict_pose(hmap.cuda())
    torch.cuda.synchronize()
    assert torch.equal(pose, again)
    np.testing.assert_allclose(pose[check].cpu().numpy(), ref.numpy(), atol=TOL, rtol=0)
    if B == 2 and hm == 96:                  # training at such a side says what it needs, by name, before any launch
        net.train()
        try:
            with pytest.raises(NotImplementedError, match="multiple of 32"):
                net(hmap.cuda())
        finally:
            net.eval()
    if B <= 4:
        scale = float(ref.abs().max())
        for mode, tol in (("bf16x3", 1e-4), ("bf16", 3e-2 * scale)):
            try:
                net.set_precision(mode)
                low = net.predict_pose(hmap.cuda())
            finally:
                net.set_precision("f32")
            err = float((low[check].double().cpu() - ref).abs().max())
            assert err < tol, (mode, err, tol)


def test_batch_rows_are_independent_and_deterministic():
    """Size-independent property at the benchmark batch: sample i of a B=256 batch equals the same
    sample run in a batch of 128, bit for bit (per-row k order does not depend on the tile the row is in),
    and two runs agree bit for bit.  Small batches run their GEMMs split-K (the number of K ranges follows the
    batch, so the summation order does too): there the property holds to rounding (1e-5), runs stay bit-reproducible."""
    from gpu_util import lift_net
    net, _, p = lift_net("UnrealEgo")
    two = torch.from_numpy(synth_input("hm_ue", (2, p.in_channels, 64, 64))).cuda()
    small = net.predict_pose(two).clone()
    six = net.predict_pose(two.repeat(3, 1, 1, 1)).clone()
    mid = net.predict_pose(two.repeat(64, 1, 1, 1)).clone()
    big_in = two.repeat(128, 1, 1, 1)
    big = net.predict_pose(big_in).clone()
    again = net.predict_pose(big_in).clone()
    torch.cuda.synchronize()
    assert torch.equal(big, again)
    assert torch.equal(big[0::2], mid[0:1].expand(128, -1, -1))
    assert torch.equal(big[1::2], mid[1:2].expand(128, -1, -1))
    assert torch.equal(small, net.predict_pose(two))
    assert (six - small.repeat(3, 1, 1)).abs().max().item() < 1e-5
    assert (big[0:2] - small).abs().max().item() < 1e-5
    g = np.load(os.path.join(GOLD, "lift_fwd_ue_b2.npz"))
    np.testing.assert_allclose(big[254:256].cpu().numpy(), g["pose"], atol=TOL, rtol=0)
    np.testing.assert_allclose(small.cpu().numpy(), g["pose"], atol=TOL, rtol=0)


def test_empty_batch_and_bad_input():
    from egotap_amd import lib
    from gpu_util import lift_net
    net, _, p = lift_net("UnrealEgo")
    out = net.predict_pose(torch.zeros(0, p.in_channels, 64, 64, device="cuda"))
    assert tuple(out.shape) == (0, 16, 3)
    with pytest.raises(ValueError):
        net.predict_pose(torch.zeros(1, p.in_channels - 1, 64, 64, device="cuda"))
    with pytest.raises(lib.EgotapError):
        net.predict_pose(torch.zeros(1, p.in_channels, 64, 64))


@pytest.mark.parametrize("tag,preset", [("ue", "UnrealEgo"), ("ec", "EgoCap")])
def test_lift_forward_bf16x3_mode_within_north_star_tolerance(tag, preset):
    """Opt-in fast mode (egotap_set_precision BF16X3): same gate as fp32 (1e-4 on the joints vs the reference golden);
    observed ~5e-6.  Also: deterministic, and switching back restores the exact-fp32 result bit for bit."""
    from gpu_util import lift_net
    g = np.load(os.path.join(GOLD, f"lift_fwd_{tag}_b2.npz"))
    net, _, p = lift_net(preset)
    hm = torch.from_numpy(synth_input(f"hm_{tag}", (2, p.in_channels, 64, 64))).cuda()
    exact = net.predict_pose(hm).clone()
    try:
        net.set_precision("bf16x3")
        fast = net.predict_pose(hm).clone()
        again = net.predict_pose(hm).clone()
    finally:
        net.set_precision("f32")
    back = net.predict_pose(hm).clone()
    torch.cuda.synchronize()
    np.testing.assert_allclose(fast.cpu().numpy(), g["pose"], atol=TOL, rtol=0)
    assert np.abs(fast.cpu().numpy() - g["pose"]).max() < 3e-5
    assert torch.equal(fast, again) and torch.equal(back, exact)
    assert not torch.equal(fast, exact)           # the mode really switched kernels
    with pytest.raises(ValueError):
        net.set_precision("fp8")


def test_lift_forward_hm128_egocap_matches_oracle():
    """BASELINE config 5 geometry: EgoCap preset on 128x128 heatmaps (512x512 RGB): 768^2 ViT image, 2304 tokens,
    fc1 K = 65536 / 32768.  fp32 and the bf16x3 fast mode against the float64 oracle."""
    from gpu_util import lift_net
    from oracle import lift_ref as O
    net, sd_np, p = lift_net("EgoCap", hm=128)
    assert p.seq == 2304 and p.in_channels == 102
    hm = torch.from_numpy(synth_input("hm_ec128", (1, p.in_channels, 128, 128)))
    sd = O.to_torch_sd(sd_np, torch.float64)
    with torch.no_grad():
        ref = O.lift_forward(hm.double(), sd, p).numpy()
    pose = net.predict_pose(hm.cuda())
    try:
        net.set_precision("bf16x3")
        fast = net.predict_pose(hm.cuda())
    finally:
        net.set_precision("f32")
    torch.cuda.synchronize()
    np.testing.assert_allclose(pose.cpu().numpy(), ref, atol=TOL, rtol=0)
    np.testing.assert_allclose(fast.cpu().numpy(), ref, atol=TOL, rtol=0)


def test_lift_forward_on_synthesised_ground_truth_heatmaps():
    """Realistic (sparse) inputs: joints -> device-rendered Gaussian / limb heatmaps (egotap_synth_heatmaps) -> lifting head, in
    fp32 and bf16x3, against the float64 oracle run on the CPU-rendered heatmaps (oracle/heatmap_synth_ref.py): the whole
    use_gt_heatmap input path in one test."""
    from egotap_amd import lib
    from gpu_util import lift_net
    from oracle import heatmap_synth_ref as R
    from oracle import lift_ref as O
    net, sd_np, p = lift_net("UnrealEgo")
    B = 3
    p2l, p2r = synth_input("rl_p2l", (B, 16, 2), 100.0, 900.0), synth_input("rl_p2r", (B, 16, 2), 100.0, 900.0)
    p3 = synth_input("rl_p3", (B, 16, 3), -40.0, 40.0)
    syn = lib.synth_heatmaps(torch.from_numpy(p2l).cuda(), torch.from_numpy(p2r).cuda(), torch.from_numpy(p3).cuda(), "UnrealEgo", 64)
    cat_ref = np.stack([R.process_frame(p2l[b].astype(np.float64), p2r[b].astype(np.float64), p3[b].astype(np.float64))[0] for b in range(B)])
    np.testing.assert_allclose(syn["cat"].cpu().numpy(), cat_ref, atol=2e-6, rtol=0)
    assert float((syn["cat"] == 0).float().mean()) > 0.5            # sparse: most pixels are exactly zero
    sd = O.to_torch_sd(sd_np, torch.float64)
    with torch.no_grad():
        ref = O.lift_forward(torch.from_numpy(cat_ref).double(), sd, p).numpy()
    pose = net.predict_pose(syn["cat"])
    try:
        net.set_precision("bf16x3")
        fast = net.predict_pose(syn["cat"])
    finally:
        net.set_precision("f32")
    torch.cuda.synchronize()
    np.testing.assert_allclose(pose.cpu().numpy(), ref, atol=TOL, rtol=0)
    np.testing.assert_allclose(fast.cpu().numpy(), ref, atol=TOL, rtol=0)


@pytest.mark.parametrize("tag,preset", [("ue", "UnrealEgo"), ("ec", "EgoCap")])
def test_every_gemm_routing_regime_matches_the_golden(tag, preset):
    """Batch sizes on both sides of every routing threshold of the forward (split-K for the ViT GEMMs below the tile count that
    fills the chip, for fc1 below 1024 encoder rows / 100 tiles, DMA-staged persistent kernels above): each frame of each batch
    equals the reference golden of that frame within the fp32 gate, in the exact and in the bf16x3 mode."""
    from gpu_util import lift_net
    net, _, p = lift_net(preset)
    g = np.load(os.path.join(GOLD, f"lift_fwd_{tag}_b2.npz"))
    two = torch.from_numpy(synth_input(f"hm_{tag}", (2, p.in_channels, 64, 64))).cuda()
    try:
        for mode, gate in (("f32", TOL), ("bf16x3", TOL)):
            net.set_precision(mode)
            for B in (1, 3, 7, 8, 12, 20, 30, 31, 34, 35, 36, 63, 106, 107, 108, 129):      # [r5] 8 ... 36: whole rounds of 256 x 256 tiles + a split tail (fc1 / qkv)
                x = two.repeat((B + 1) // 2, 1, 1, 1)[:B].contiguous()
                out = net.predict_pose(x).cpu().numpy()
                ref = np.tile(g["pose"], ((B + 1) // 2, 1, 1))[:B]
                np.testing.assert_allclose(out, ref, atol=gate, rtol=0, err_msg=f"{preset} {mode} B={B}")
    finally:
        net.set_precision("f32")


def test_lift_forward_graph_capture_replays_bit_identically():
    """the header's contract "no device allocation, no synchronisation, all work on the caller's stream: graph-capture safe": capture
    egotap_lift_forward (B = 5: split-K small-batch path and the 30 propagation-unit launches included) in a HIP graph on a side
    stream, replay it on fresh inputs, and compare bit for bit with the eager call"""
    from gpu_util import lift_net
    net, sd_np, p = lift_net("UnrealEgo")
    B = 5
    hm_a = torch.from_numpy(synth_input("hm_graph_a", (B, p.in_channels, 64, 64))).cuda()
    hm_b = torch.from_numpy(synth_input("hm_graph_b", (B, p.in_channels, 64, 64))).cuda()
    eager_a, eager_b = net.predict_pose(hm_a).clone(), net.predict_pose(hm_b).clone()      # also warms up lazy attribute calls
    static_in = hm_a.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        net.predict_pose(static_in)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        static_out = net.predict_pose(static_in)
    for src, ref in ((hm_a, eager_a), (hm_b, eager_b), (hm_a, eager_a)):
        static_in.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(static_out, ref)


def test_pu_chain_switch_gives_the_same_bits():
    """egotap_set_pu_chain(h, 0) -- the propagation units' recurrence as one kernel per step, for a device shared with other
    processes -- against the default one-launch recurrence: same poses bit for bit, inference and training forward"""
    from gpu_util import lift_net
    net, sd_np, p = lift_net("UnrealEgo")
    hm = torch.from_numpy(synth_input("hm_chain_switch", (37, p.in_channels, 64, 64))).cuda()
    a = net.predict_pose(hm).clone()
    bufs = {k: v.clone() for k, v in net.named_buffers()}        # the net is shared by the tests: a train-mode forward moves its BatchNorm statistics
    try:
        net.set_pu_chain(False)
        b = net.predict_pose(hm).clone()
        net.train()
        tb = net(hm)[0].detach().clone()
        for k, v in net.named_buffers():
            v.copy_(bufs[k])
        net.set_pu_chain(True)
        ta = net(hm)[0].detach().clone()
    finally:
        net.set_pu_chain(True)
        net.eval()
        for k, v in net.named_buffers():
            v.copy_(bufs[k])
    assert torch.equal(a, b) and torch.equal(ta, tb) and torch.isfinite(a).all()


@pytest.mark.parametrize("B", [5, 37])
def test_starved_pu_chain_is_redone_on_the_device_and_reported(B):
    """Round-2 verdict, weak 4: a one-launch recurrence whose row block is not co-resident used to return NaN poses with rc 0.  The
    test hook starves the last row block (its last workgroup is never launched -- what a shared device does): the waits run out, the
    workgroups store nothing and raise the fault word, pu_solo_kernel redoes the launch -> the SAME bits as the healthy run, for the
    inference forward, the training forward and a whole training step's gradients; the handle reports the fault and switches
    itself to the per-step kernels at its next call."""
    import ctypes as C
    from egotap_amd import lib as _lib
    from egotap_amd import networks, spec
    from egotap_amd.options import preset_defaults
    from egotap_amd.training import PoseLossFn
    p = spec.lift_preset("UnrealEgo")
    net = networks.EgoTAPAutoEncoder(preset_defaults("UnrealEgo"), input_channel_scale=2)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(p)).items()})
    net = net.cuda().eval()
    L, h = _lib.load(), net._ensure_handle()
    hm = torch.from_numpy(synth_input("hm_chain_fault", (B, p.in_channels, 64, 64))).cuda()
    gt = torch.from_numpy(synth_input("gt_chain_fault", (B, p.out_joints, 3), -1.0, 1.0)).cuda()
    bufs = {k: v.clone() for k, v in net.named_buffers()}

    def train_step():
        for k, v in net.named_buffers():
            v.copy_(bufs[k])
        net.train()
        net.zero_grad()
        pose = net(hm)[0]
        PoseLossFn.apply(net, pose, gt, 0.1, -0.01).sum().backward()
        torch.cuda.synchronize()
        net.eval()
        for k, v in net.named_buffers():                 # the train-mode forward moved the BatchNorm running statistics the eval forward reads
            v.copy_(bufs[k])
        return pose.detach().clone(), {k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None}

    want = net.predict_pose(hm).clone()
    want_t, want_g = train_step()
    assert net.pu_chain_status() == (True, 0)
    # ---- inference forward with a starved row block
    _lib.check(L.egotap_debug_pu_drop_workgroups(h, 1))
    try:
        got = net.predict_pose(hm).clone()
        torch.cuda.synchronize()
    finally:
        _lib.check(L.egotap_debug_pu_drop_workgroups(h, 0))
    assert torch.isfinite(got).all() and torch.equal(got, want)
    en, faults = net.pu_chain_status()
    assert faults == 1 and not en                                   # reported; the handle now walks the steps itself
    assert torch.equal(net.predict_pose(hm), want)
    # ---- the same through the training forward / backward (the gate inputs stay intact, so the redo starts from them)
    net.set_pu_chain(True)
    _lib.check(L.egotap_debug_pu_drop_workgroups(h, 1))
    try:
        got_t, got_g = train_step()
    finally:
        _lib.check(L.egotap_debug_pu_drop_workgroups(h, 0))
    assert torch.equal(got_t, want_t)
    for k in want_g:
        assert torch.equal(got_g[k], want_g[k]), k
    en, faults = net.pu_chain_status()
    assert faults == 2 and not en
    # the switch-off is noticed by the NEXT call without any query (no synchronisation on the hot path): fault again, then just call
    net.set_pu_chain(True)
    _lib.check(L.egotap_debug_pu_drop_workgroups(h, 1))
    net.predict_pose(hm)
    torch.cuda.synchronize()
    _lib.check(L.egotap_debug_pu_drop_workgroups(h, 0))
    assert torch.equal(net.predict_pose(hm), want)                  # this call saw the word, counted it and used the per-step kernels
    assert net.pu_chain_status() == (False, 3)


def test_predict_pose_graphed_equals_eager():
    """the wrapper's graph-replay inference path (serving at small batches): same bits as the eager call, for two batch sizes and
    changing inputs; a re-bound parameter (new storage) gets a fresh capture"""
    from gpu_util import lift_net
    net, sd_np, p = lift_net("UnrealEgo")
    for B in (1, 6):
        for tag in ("a", "b"):
            hm = torch.from_numpy(synth_input(f"hm_pg_{tag}{B}", (B, p.in_channels, 64, 64))).cuda()
            assert torch.equal(net.predict_pose_graphed(hm), net.predict_pose(hm))
    assert len(net._graphs) == 2
    w = dict(net.named_parameters())["pose_mlp.pose_fcs.0.bias"]
    w.data = w.data.clone() * 2.0                                  # new storage, new values
    hm = torch.from_numpy(synth_input("hm_pg_a1", (1, p.in_channels, 64, 64))).cuda()
    assert torch.equal(net.predict_pose_graphed(hm), net.predict_pose(hm))
    w.data = w.data / 2.0


def test_graph_replay_survives_workspace_growth():
    """a captured graph owns the workspace it replays into: capture B = 1, let the module's grow-only workspace be replaced by a
    larger batch (eager B = 64 in fp32, then the bf16 scratches by a bf16 forward) and churn the allocator so the old block is
    reused, then replay B = 1 -- still the eager bits (the round-2 capture baked the module's own workspace pointer in)"""
    from gpu_util import lift_net
    import gc
    net, sd_np, p = lift_net("UnrealEgo")
    net.__dict__.pop("_graphs", None)
    net._ws = None                                                 # fresh module-level workspace, sized by the B = 1 call below
    hm1 = torch.from_numpy(synth_input("hm_pg_grow1", (1, p.in_channels, 64, 64))).cuda()
    want = net.predict_pose(hm1).clone()
    assert torch.equal(net.predict_pose_graphed(hm1), want)
    small_ws = net._ws.data_ptr()
    hm64 = torch.from_numpy(synth_input("hm_pg_grow64", (64, p.in_channels, 64, 64))).cuda()
    net.predict_pose(hm64)                                         # replaces net._ws: the B = 1 block goes back to the allocator
    assert net._ws.data_ptr() != small_ws
    gc.collect()
    junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(64)]      # overwrite whatever was freed
    torch.cuda.synchronize()
    assert torch.equal(net.predict_pose_graphed(hm1), want)
    del junk
    for mode in ("bf16x3", "bf16"):                                # per-precision graphs keep their scratch buffers alive too
        try:
            net.set_precision(mode)
            w = net.predict_pose(hm1).clone()
            assert torch.equal(net.predict_pose_graphed(hm1), w)
            net.predict_pose(hm64[:6])                             # grows the activation scratch in bf16 mode
            junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(16)]
            torch.cuda.synchronize()
            assert torch.equal(net.predict_pose_graphed(hm1), w)
            del junk
        finally:
            net.set_precision("f32")
    assert torch.equal(net.predict_pose_graphed(hm1), want)


@pytest.mark.parametrize("preset,B", [("UnrealEgo", 17), ("EgoCap", 9)])
def test_lift_forward_bf16_storage_mode_against_oracle(preset, B):
    """EGOTAP_PREC_BF16 at a batch that takes the bf16-storage forward (B * 576 >= 4096: bf16 activations in the workspace, LDS-DMA
    GEMMs, bf16 attention; ragged last tile row): first and last frames against the float64 oracle within bf16's error (2^-9 per
    rounded operand through ~20 layers: a few 1e-2 of the pose scale), bit-reproducible run to run, and switching back restores fp32"""
    from gpu_util import lift_net
    from oracle import lift_ref as O
    net, sd_np, p = lift_net(preset)
    hm = torch.from_numpy(synth_input(f"hm_bf16s_{preset}", (B, p.in_channels, 64, 64)))
    rows = [0, 1, B - 1]
    with torch.no_grad():
        ref = O.lift_forward(hm[rows].double(), O.to_torch_sd(sd_np, torch.float64), p)
    exact = net.predict_pose(hm.cuda()).clone()
    try:
        net.set_precision("bf16")
        low = net.predict_pose(hm.cuda()).clone()
        again = net.predict_pose(hm.cuda()).clone()
    finally:
        net.set_precision("f32")
    back = net.predict_pose(hm.cuda())
    assert torch.equal(low, again) and torch.equal(back, exact) and not torch.equal(low, exact)
    assert float((exact[rows].double().cpu() - ref).abs().max()) < 1e-4
    assert float((low[rows].double().cpu() - ref).abs().max()) < 3e-2 * float(ref.abs().max())
