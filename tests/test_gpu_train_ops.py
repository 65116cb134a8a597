"""GPU parity of the training-step operators (C ABI) against float64 torch autograd on the CPU."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g, dtype=torch.float64) * (hi - lo) + lo).float()


def _close(got, ref64, atol, rtol=1e-4, msg=""):
    got = got.detach().cpu().double()
    err = (got - ref64).abs()
    tol = atol + rtol * ref64.abs()
    assert bool((err <= tol).all()), f"{msg} max err {err.max().item():.3e} (max ref {ref64.abs().max().item():.3e})"


def _handle():
    from gpu_util import lift_net
    net, _, p = lift_net("UnrealEgo")
    net._bind(torch.device("cuda", 0))
    return net._ensure_handle(), net, p


@pytest.mark.parametrize("M,N,K", [(700, 256, 256), (5000, 512, 2048), (300, 128, 512), (9000, 1024, 1024), (60, 768, 256)])
def test_gemm_tn(M, N, K):
    from egotap_amd import train_ops as T
    h, _, _ = _handle()
    dy, x = _rand((M, N), 1), _rand((M, K), 2)
    dw = torch.full((N, K), 3.0, device="cuda")
    T.gemm_tn(h, dy.cuda(), x.cuda(), dw, M, N, K)
    ref = dy.double().T @ x.double()
    _close(dw, ref, atol=2e-5 * math.sqrt(M))
    T.gemm_tn(h, dy.cuda(), x.cuda(), dw, M, N, K, accumulate=True)
    _close(dw, 2 * ref, atol=4e-5 * math.sqrt(M))


@pytest.mark.parametrize("M,N,K,strided", [(1024, 256, 256, False), (4128, 512, 1024, False), (2080, 256, 512, True), (8192, 1024, 256, False),
                                             (700, 256, 256, False), (1200, 256, 512, True)])
def test_gemm_tn_bias_one_call(M, N, K, strided):
    """[r3] weight + bias gradient in one call (egotap_train_gemm_tn_bias).  M % 32 == 0 and M >= 1024: the DMA-staged kernel, whose first k-tile
    column of workgroups sums the dY rows it stages; the last two shapes fall back to the two operators.  Against float64; accumulate; the
    same bits on a second run (fixed summation order); the plain weight-gradient entry takes the same kernel and gives the same dW."""
    from egotap_amd import train_ops as T
    h, _, _ = _handle()
    ld = 3 * N if strided else N
    dyf, x = _rand((M, ld), 11), _rand((M, K), 12)
    dyc = dyf.cuda()
    dy_view = dyc[:, N:] if strided else dyc
    dy_ref = (dyf[:, N:2 * N] if strided else dyf).double()
    dw, db = torch.full((N, K), 3.0, device="cuda"), torch.full((N,), -2.0, device="cuda")
    T.gemm_tn_bias(h, dy_view, x.cuda(), dw, db, M, N, K, ldy=ld if strided else 0)
    ref_w, ref_b = dy_ref.T @ x.double(), dy_ref.sum(0)
    _close(dw, ref_w, atol=2e-5 * math.sqrt(M), msg="dw")
    _close(db, ref_b, atol=2e-5 * math.sqrt(M), msg="db")
    dw2, db2 = torch.empty_like(dw), torch.empty_like(db)
    T.gemm_tn_bias(h, dy_view, x.cuda(), dw2, db2, M, N, K, ldy=ld if strided else 0)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    dw3 = torch.empty_like(dw)
    T.gemm_tn(h, dy_view, x.cuda(), dw3, M, N, K, ldy=ld if strided else 0)
    assert torch.equal(dw, dw3)
    T.gemm_tn_bias(h, dy_view, x.cuda(), dw, db, M, N, K, accumulate=True, ldy=ld if strided else 0)
    _close(dw, 2 * ref_w, atol=4e-5 * math.sqrt(M), msg="dw accumulate")
    _close(db, 2 * ref_b, atol=4e-5 * math.sqrt(M), msg="db accumulate")


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
def test_gemm_tn_bias_bf16_modes(mode):
    """[r3] the same call in the bf16 product modes: dW on the bf16 matrix cores (hi + lo split or plain rounding), db summed from the fp32 dY values
    the kernel stages (before the rounding: fp32-grade in both modes); equal to the two separate operators' dW bit for bit"""
    from egotap_amd import lib as L
    from egotap_amd import train_ops as T
    h, _, _ = _handle()
    M, N, K = 2048, 512, 256
    dyf, x = _rand((M, 3 * N), 21), _rand((M, K), 22)
    dyc = dyf.cuda()
    L.check(L.load().egotap_set_precision(h, L.PRECISIONS[mode]))
    try:
        dw, db = torch.empty((N, K), device="cuda"), torch.empty((N,), device="cuda")
        T.gemm_tn_bias(h, dyc[:, N:], x.cuda(), dw, db, M, N, K, ldy=3 * N)
        dw2 = torch.empty_like(dw)
        T.gemm_tn(h, dyc[:, N:], x.cuda(), dw2, M, N, K, ldy=3 * N)
        torch.cuda.synchronize()
    finally:
        L.check(L.load().egotap_set_precision(h, L.PRECISIONS["f32"]))
    ref_w, ref_b = dyf[:, N:2 * N].double().T @ x.double(), dyf[:, N:2 * N].double().sum(0)
    _close(db, ref_b, atol=2e-5 * math.sqrt(M), msg="db")
    _close(dw, ref_w, atol=(2e-4 if mode == "bf16x3" else 2e-2) * math.sqrt(M), msg="dw")
    assert torch.equal(dw, dw2)


def test_gemm_tn_strided_dy_and_colsum_transpose():
    from egotap_amd import train_ops as T
    h, _, _ = _handle()
    M, N, K = 1200, 256, 512
    dy3, x = _rand((M, 3 * N), 3), _rand((M, K), 4)
    dy3c = dy3.cuda()
    dw = torch.empty((N, K), device="cuda")
    T.gemm_tn(h, dy3c[:, N:], x.cuda(), dw, M, N, K, ldy=3 * N)
    _close(dw, dy3[:, N:2 * N].double().T @ x.double(), atol=1e-3)
    out = torch.empty(N, device="cuda")
    T.colsum(dy3c[:, 2 * N:], out, M, N, ldy=3 * N)
    _close(out, dy3[:, 2 * N:].double().sum(0), atol=1e-3)
    w = _rand((300, 130), 5)
    _close(T.transpose(w.cuda()), w.double().T, atol=0)
    wide = torch.zeros((130, 900), device="cuda")
    T.transpose(w.cuda(), out=wide[:, 300:], ldo=900)
    assert torch.equal(wide[:, 300:600].cpu(), w.T) and float(wide[:, :300].abs().max()) == 0


def test_gemm_nt_training_epilogues():
    from egotap_amd import train_ops as T
    h, _, _ = _handle()
    M, N, K = 1500, 512, 256
    x, w, b, r = _rand((M, K), 6), _rand((N, K), 7, -0.1, 0.1), _rand((N,), 8), _rand((M, N), 9)
    xd, wd = x.double(), w.double()
    z_ref = xd @ wd.T + b.double()
    _close(T.gemm_nt(h, x.cuda(), w.cuda(), None, M, N, K, epi=T.TE_NONE), xd @ wd.T, 2e-5)
    _close(T.gemm_nt(h, x.cuda(), w.cuda(), None, M, N, K, epi=T.TE_ACCUM, r=r.cuda()), xd @ wd.T + r.double(), 2e-5)
    zbuf = torch.full((M + 300, N), 7.0, device="cuda")        # canary rows behind z: the ragged last tile must not touch them
    z = zbuf[:M]
    hcu = T.gemm_nt(h, x.cuda(), w.cuda(), b.cuda(), M, N, K, epi=T.TE_BIAS_GELU_SAVE, z=z)
    _close(z, z_ref, 2e-5)
    assert float((zbuf[M:] - 7.0).abs().max()) == 0.0
    _close(hcu, 0.5 * z_ref * (1 + torch.erf(z_ref / math.sqrt(2))), 2e-5)
    dg = 0.5 * (1 + torch.erf(r.double() / math.sqrt(2))) + r.double() * torch.exp(-0.5 * r.double() ** 2) / math.sqrt(2 * math.pi)
    _close(T.gemm_nt(h, x.cuda(), w.cuda(), None, M, N, K, epi=T.TE_GELU_GRAD, r=r.cuda()), (xd @ wd.T) * dg, 3e-5)
    # strided A (a column block of a wider matrix)
    xw = _rand((M, 3 * K), 10)
    _close(T.gemm_nt(h, xw.cuda()[:, K:], w.cuda(), None, M, N, K, epi=T.TE_NONE, lda=3 * K), xw[:, K:2 * K].double() @ wd.T, 2e-5)


def test_layernorm_fwd_bwd():
    from egotap_amd import train_ops as T
    rows = 777
    x, g, b, dy, dres = _rand((rows, 1024), 11, -2, 2), _rand((1024,), 12, 0.5, 1.5), _rand((1024,), 13), _rand((rows, 1024), 14), _rand((rows, 1024), 15)
    xr, gr, br = x.double().requires_grad_(True), g.double().requires_grad_(True), b.double().requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(xr, (1024,), gr, br, 1e-12)
    y_ref.backward(dy.double())
    y, mean, rstd = T.layernorm_fwd(x.cuda(), g.cuda(), b.cuda())
    _close(y, y_ref.detach(), 5e-6)
    dgam, dbet = torch.empty(1024, device="cuda"), torch.empty(1024, device="cuda")
    dx = T.layernorm_bwd(x.cuda(), dy.cuda(), g.cuda(), mean, rstd, dgam, dbet, dres=dres.cuda())
    _close(dx, xr.grad + dres.double(), 2e-5)
    _close(dgam, gr.grad, 2e-4)
    _close(dbet, br.grad, 2e-4)


@pytest.mark.parametrize("R,Cc", [(60, 128), (7680, 512), (1000, 2048)])
def test_bn_lrelu_fwd_bwd(R, Cc):
    from egotap_amd import train_ops as T
    z, g, b, dy = _rand((R, Cc), 21, -2, 2), _rand((Cc,), 22, 0.5, 1.5), _rand((Cc,), 23), _rand((R, Cc), 24)
    rm, rv = _rand((Cc,), 25, -0.1, 0.1), _rand((Cc,), 26, 0.5, 1.5)
    zr, gr, br = z.double().requires_grad_(True), g.double().requires_grad_(True), b.double().requires_grad_(True)
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    y_ref = torch.nn.functional.leaky_relu(torch.nn.functional.batch_norm(zr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5), 0.2)
    y_ref.backward(dy.double())
    rmc, rvc = rm.cuda(), rv.cuda()
    y, mean, rstd = T.bn_lrelu_fwd(z.cuda(), g.cuda(), b.cuda(), rmc, rvc)
    _close(y, y_ref.detach(), 1e-5)
    _close(rmc, rm_ref, 1e-6)
    _close(rvc, rv_ref, 1e-5)
    dgam, dbet = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
    dz = T.bn_lrelu_bwd(z.cuda(), y, dy.cuda(), g.cuda(), mean, rstd, dgam, dbet)
    _close(dz, zr.grad, 2e-5, rtol=1e-3)
    _close(dgam, gr.grad, 1e-3, rtol=1e-3)
    _close(dbet, br.grad, 1e-3, rtol=1e-3)


@pytest.mark.parametrize("B,N,heads", [(1, 64, 1), (2, 576, 8), (1, 96, 3)])
def test_attention_fwd_bwd(B, N, heads):
    from egotap_amd import train_ops as T
    D = heads * 128
    qkv, dctx = _rand((B * N, 3 * D), 31, -1.5, 1.5), _rand((B * N, D), 32)
    qkvr = qkv.double().requires_grad_(True)
    q, k, v = [t.reshape(B, N, heads, 128).transpose(1, 2) for t in qkvr.split(D, dim=1)]
    s = q @ k.transpose(-1, -2) / math.sqrt(128.0)
    ctx_ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * N, D)
    ctx_ref.backward(dctx.double())
    ctx, lse = T.attention_fwd(qkv.cuda(), B, N, heads)
    _close(ctx, ctx_ref.detach(), 5e-6)
    _close(lse.reshape(B, heads, N), torch.logsumexp(s.detach(), -1), 1e-5)
    dqkv = T.attention_bwd(qkv.cuda(), ctx, dctx.cuda(), lse, B, N, heads)
    _close(dqkv, qkvr.grad, 1e-5, rtol=1e-3)


@pytest.mark.parametrize("preset", ["UnrealEgo", "EgoCap"])
def test_pose_loss(preset):
    from egotap_amd import train_ops as T
    from gpu_util import lift_net
    from oracle import lift_ref as O
    net, _, p = lift_net(preset)
    J = p.out_joints
    pred, gt = _rand((7, J, 3), 41, -20, 20), _rand((7, J, 3), 42, -20, 20)
    pr = pred.double().requires_grad_(True)
    lp, lc = O.loss_mpjpe(pr, gt.double()) * 0.1, O.loss_cos_sim(pr, gt.double(), p) * (-0.01) * 0.1
    g_pose, = torch.autograd.grad(lp, pr, retain_graph=True)
    g_cos, = torch.autograd.grad(lc, pr)
    out, dpred = T.pose_loss(net._ensure_handle(), pred.cuda(), gt.cuda())
    np.testing.assert_allclose(out.cpu().numpy(), [lp.item(), lc.item()], rtol=2e-5, atol=1e-7)
    _close(dpred[0], g_pose, 1e-7, rtol=1e-4)          # the two terms are kept apart (PoseLossFn weighs them with the upstream gradients)
    _close(dpred[1], g_cos, 1e-7, rtol=1e-4)
    # any weighting of the two returned losses gets its exact gradient through the autograd wrapper
    from egotap_amd.training import PoseLossFn
    pc = pred.cuda().requires_grad_(True)
    both = PoseLossFn.apply(net, pc, gt.cuda(), 0.1, -0.01)
    (2.0 * both[0] - 3.0 * both[1]).backward()
    _close(pc.grad, 2.0 * g_pose - 3.0 * g_cos, 1e-7, rtol=1e-4)


def test_adamw_matches_torch():
    from egotap_amd import train_ops as T
    n = 10007
    p0, g1, g2 = _rand((n,), 51), _rand((n,), 52, -0.1, 0.1), _rand((n,), 53, -0.1, 0.1)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, eps=1e-4, weight_decay=0.02)
    p, m, v = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step, g in enumerate((g1, g2), start=1):
        ref.grad = g.clone()
        opt.step()
        T.adamw(p, g.cuda(), m, v, 1e-3, step, eps=1e-4, weight_decay=0.02)
    _close(p, ref.detach().double(), 1e-6)


def test_adamw_multi_tensor_launch_matches_torch_on_ragged_segments():
    """egotap_train_adamw_multi (one launch for every tensor whose gradient lives in one arena; [r5] four elements per thread, 16-byte accesses
    inside a segment): tensors of 3 / 1 / 1024 / 7 / 65 / 4099 / 30000 elements at 64-element-aligned arena offsets (padding between them), one
    PARAMETER that is a 4-byte-aligned view (element-wise path), two steps against torch.optim.AdamW; padding and neighbours untouched."""
    from egotap_amd.training import EgotapAdamW
    sizes = [3, 1, 1024, 7, 65, 4099, 30000, 130]
    offs, o = [], 0
    for n in sizes:
        offs.append(o)
        o += (n + 63) // 64 * 64
    arena = torch.zeros(o, device="cuda")
    backing = torch.zeros(sum(sizes) + 16, device="cuda")          # parameter 5 is a view at an odd element offset: 4-byte aligned only
    ps, refs = [], []
    for i, n in enumerate(sizes):
        init = _rand((n,), 300 + i)
        if i == 5:
            view = backing[1:1 + n]
            view.copy_(init.cuda())
            prm = torch.nn.Parameter(view)
            assert prm.data_ptr() % 16 != 0
        else:
            prm = torch.nn.Parameter(init.cuda())
        prm.grad = arena[offs[i]:offs[i] + n]
        ps.append(prm)
        refs.append(torch.nn.Parameter(init.clone()))
    opt = EgotapAdamW(ps, lr=2e-3, eps=1e-4, weight_decay=0.01)
    ref_opt = torch.optim.AdamW(refs, lr=2e-3, eps=1e-4, weight_decay=0.01)
    for step in range(2):
        arena.fill_(float("nan"))                                   # the padding between segments is never read into an update
        for i, n in enumerate(sizes):
            gr = _rand((n,), 400 + 10 * step + i, -0.1, 0.1)
            arena[offs[i]:offs[i] + n] = gr.cuda()
            refs[i].grad = gr.clone()
        opt.step()
        ref_opt.step()
    torch.cuda.synchronize()
    assert opt._flat, "the arena launch was not taken"
    for i in range(len(sizes)):
        _close(ps[i].detach(), refs[i].detach().double(), 1e-6)
    assert float(backing[0]) == 0.0 and float(backing[1 + sizes[5]:].abs().max()) == 0.0      # the view's neighbours


def test_pu_chain_and_pose_head_backward():
    """SkelNet(PU) + pose head: forward in training mode and full backward vs autograd of the oracle"""
    from egotap_amd import lib as L, train_ops as T
    from oracle import lift_ref as O
    h, net, p = _handle()
    B, J, hid, H = 3, p.n_joints_hm, p.hidden, p.pu_hidden
    posz, rotz, dpose = _rand((B * 2 * J, hid), 61), _rand((B * 2 * J, hid), 62), _rand((B, 16, 3), 63)
    sd64 = {k: v.detach().cpu().double() for k, v in net.state_dict().items() if v.is_floating_point()}
    keys = [k for k in sd64 if k.startswith("skel_sequential_layer.") or k.startswith("pose_mlp.") or k.startswith("global_mlp.")]
    leaves = {k: sd64[k].clone().requires_grad_(True) for k in keys}
    sdr = dict(sd64); sdr.update(leaves)
    pz, rz = posz.double().requires_grad_(True), rotz.double().requires_grad_(True)
    pos_j, rot_j = O.stereo_interleave(pz, B, p), O.stereo_interleave(rz, B, p)
    skel = O.pu_chain(pos_j.transpose(0, 1), rot_j.transpose(0, 1), sdr)
    pose_ref = O.pose_head(pos_j, skel, sdr, p)
    pose_ref.backward(dpose.double())
    # GPU
    lib = L.load()
    nb, off = C.c_size_t(), C.c_size_t()
    L.check(lib.egotap_train_pu_saved_bytes(h, B, C.byref(nb), C.byref(off)))
    saved = torch.empty(nb.value, dtype=torch.uint8, device="cuda")
    pc, rc = posz.cuda(), rotz.cuda()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.egotap_train_pu_fwd(h, T._p(pc), T._p(rc), B, T._p(saved), saved.numel(), st))
    hs1 = saved[off.value: off.value + 4 * J * B * H].view(torch.float32)
    _close(hs1.reshape(J, B, H), skel.detach(), 2e-6)
    pose = torch.empty((B, 16, 3), device="cuda")
    L.check(lib.egotap_train_pose_head_fwd(h, T._p(pc), T._p(hs1), B, T._p(pose), st))
    _close(pose, pose_ref.detach(), 3e-6)
    dposz, dhs1, drotz = torch.empty_like(pc), torch.empty(J * B * H, device="cuda"), torch.empty_like(rc)
    g = {k: torch.zeros_like(net.state_dict()[k]) for k in keys}
    dpose_dev = dpose.cuda()                  # (kept alive: the call takes a raw pointer)
    L.check(lib.egotap_train_pose_head_bwd(h, T._p(pc), T._p(hs1), T._p(dpose_dev), B, T._p(dposz), T._p(dhs1),
                                           T._p(g["pose_mlp.pose_fcs.0.weight"]), T._p(g["pose_mlp.pose_fcs.0.bias"]),
                                           T._p(g["global_mlp.pose_fcs.0.weight"]), T._p(g["global_mlp.pose_fcs.0.bias"]), 0, st))
    wsb = C.c_size_t()
    L.check(lib.egotap_train_pu_bwd_ws_bytes(h, B, C.byref(wsb)))
    ws = torch.empty(wsb.value, dtype=torch.uint8, device="cuda")
    order = ["0.x2f", "0.x2h", "0.b2h", "0.h2h", "1.x2f", "1.x2h", "1.h2h"]
    ptrs = (C.c_void_p * 14)()
    for i, nme in enumerate(order):
        ptrs[2 * i] = g[f"skel_sequential_layer.lstm_custom.layers.{nme}.weight"].data_ptr()
        ptrs[2 * i + 1] = g[f"skel_sequential_layer.lstm_custom.layers.{nme}.bias"].data_ptr()
    L.check(lib.egotap_train_pu_bwd(h, T._p(pc), T._p(rc), B, T._p(saved), T._p(dhs1), T._p(dposz), T._p(drotz), ptrs, 0, T._p(ws),
                                    ws.numel(), st))
    torch.cuda.synchronize()
    _close(dposz, pz.grad, 2e-6, rtol=1e-3, msg="dposz")
    _close(drotz, rz.grad, 2e-6, rtol=1e-3, msg="drotz")
    for k in keys:
        _close(g[k], leaves[k].grad, 3e-6, rtol=2e-3, msg=k)


# ---- weight-gradient / training GEMMs on the bf16 matrix cores (gemm_tn_bf16.h, gemm_bf16.h) selected by the handle's precision
@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
@pytest.mark.parametrize("M,N,K", [(1024, 256, 256), (5000, 512, 1024), (9001, 1024, 256)])
def test_gemm_tn_bf16_modes(mode, M, N, K):
    """transposed LDS reads (ds_read_b64_tr_b16): asymmetric random operands, ragged M (zero-filled tail rows), split-M slabs.
    bf16x3 against float64 with the 2^-16-per-product error model; bf16 against the float64 product of the rounded operands."""
    from egotap_amd import train_ops as T
    h, net, _ = _handle()
    dy, x = _rand((M, N), 61), _rand((M, K), 62)
    dw = torch.full((N, K), 3.0, device="cuda")
    try:
        net.set_precision(mode)
        T.gemm_tn(h, dy.cuda(), x.cuda(), dw, M, N, K)
        first = dw.clone()
        T.gemm_tn(h, dy.cuda(), x.cuda(), dw, M, N, K, accumulate=True)
        again = torch.empty_like(dw)
        T.gemm_tn(h, dy.cuda(), x.cuda(), again, M, N, K)
    finally:
        net.set_precision("f32")
    assert torch.equal(first, again)                       # fixed summation order
    if mode == "bf16x3":
        ref = dy.double().T @ x.double()
        scale = float((dy.double().abs().T @ x.double().abs()).mean())
        err = (first.cpu().double() - ref).abs()
        assert float(err.max()) < 2.0 ** -15 * scale and float(err.pow(2).mean().sqrt()) < 2.0 ** -17 * scale
        _close(dw, 2 * ref, atol=2.0 ** -14 * scale)
    else:
        ref = dy.bfloat16().double().T @ x.bfloat16().double()
        _close(first, ref, atol=4e-5 * math.sqrt(M))


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
def test_gemm_nt_training_epilogues_bf16_modes(mode):
    """forward / input-gradient GEMMs of the training step follow the handle's precision (GELU-save and GELU-grad epilogues)"""
    from egotap_amd import train_ops as T
    h, net, _ = _handle()
    M, N, K = 1100, 512, 256
    x, w, b = _rand((M, K), 71), _rand((N, K), 72, -0.1, 0.1), _rand((N,), 73)
    xr, wr = (x, w) if mode == "bf16x3" else (x.bfloat16().float(), w.bfloat16().float())
    zref = xr.double() @ wr.double().T + b.double()
    tol = 1e-4 if mode == "bf16x3" else 2e-5
    try:
        net.set_precision(mode)
        z = torch.empty((M, N), device="cuda")
        y = T.gemm_nt(h, x.cuda(), w.cuda(), b.cuda(), M, N, K, epi=T.TE_BIAS_GELU_SAVE, z=z)
        d = T.gemm_nt(h, x.cuda(), w.cuda(), None, M, N, K, epi=T.TE_GELU_GRAD, r=z)
    finally:
        net.set_precision("f32")
    _close(z, zref, atol=tol)
    _close(y, 0.5 * zref * (1.0 + torch.erf(zref / math.sqrt(2.0))), atol=tol)
    zz = z.cpu().double()
    dg = 0.5 * (1.0 + torch.erf(zz / math.sqrt(2.0))) + zz * torch.exp(-0.5 * zz * zz) / math.sqrt(2.0 * math.pi)
    _close(d, (xr.double() @ wr.double().T) * dg, atol=tol)


@pytest.mark.parametrize("mode,tol", [("bf16x3", 5e-5), ("bf16", 3e-2)])
@pytest.mark.parametrize("B,N,heads", [(1, 64, 1), (2, 576, 8), (1, 96, 3)])
def test_attention_fwd_bwd_bf16_modes(mode, tol, B, N, heads):
    """attention forward (with log-sum-exp) and the three backward kernels on the bf16 matrix cores: row images, transposed
    reads and split probability / dS accumulators against float64 autograd; asymmetric data, partial last group (N = 96)."""
    from egotap_amd import train_ops as T
    D = heads * 128
    qkv, dctx = _rand((B * N, 3 * D), 31, -1.5, 1.5), _rand((B * N, D), 32)
    qkvr = qkv.double().requires_grad_(True)
    q, k, v = [t.reshape(B, N, heads, 128).transpose(1, 2) for t in qkvr.split(D, dim=1)]
    s = q @ k.transpose(-1, -2) / math.sqrt(128.0)
    ctx_ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * N, D)
    ctx_ref.backward(dctx.double())
    ctx, lse = T.attention_fwd(qkv.cuda(), B, N, heads, mode)
    _close(ctx, ctx_ref.detach(), tol)
    _close(lse.reshape(B, heads, N), torch.logsumexp(s.detach(), -1), tol)
    dqkv = T.attention_bwd(qkv.cuda(), ctx, dctx.cuda(), lse, B, N, heads, mode)
    _close(dqkv, qkvr.grad, tol, rtol=1e-3 if mode == "bf16x3" else 5e-2)
    assert torch.equal(dqkv, T.attention_bwd(qkv.cuda(), ctx, dctx.cuda(), lse, B, N, heads, mode))
