"""GPU parity of the bf16-STORAGE operators (EGOTAP_PREC_BF16 with bf16 tensors in HBM; csrc/gemm_bf16s.h, gemm_tn_bf16s.h, ...),
each through the C ABI against a float64 product of the SAME bf16 operands: the only differences left are the fp32 accumulation
order and the final rounding of a bf16 output (2^-9 relative)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * (hi - lo) + lo).float()


def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _dgelu(z):
    return 0.5 * (1.0 + torch.erf(z / math.sqrt(2.0))) + z * torch.exp(-0.5 * z * z) / math.sqrt(2.0 * math.pi)


def _close_bf16(got, ref, what, rel=2.0 ** -8, abs_=1e-5):
    """a bf16 output is the fp32 result rounded to 8 significant bits: |err| <= 2^-9 |ref| + accumulation noise"""
    got, ref = got.double().cpu(), ref.double()
    err = (got - ref).abs()
    lim = rel * ref.abs() + abs_
    bad = err > lim
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} of {bad.numel()} outside 2^-8 |ref| + {abs_}; worst {float((err - lim).max()):.3e}"


SHAPES = [(256, 256, 32), (300, 512, 96), (257, 256, 128), (1153, 768, 1024), (5000, 256, 64), (2304, 1024, 4096), (70000, 1024, 256)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_nt_bf16_out(M, N, K):
    """epi 0: out bf16 = x w^T + b -- fewer K-tiles than ring stages, ragged M, tiles < CUs, several tiles per workgroup; an
    asymmetric weight catches a transposed fragment, canary rows catch stores past M"""
    from egotap_amd import bf16s
    x, w, b = _rand((M, K), 1).bfloat16(), (_rand((N, K), 2) / math.sqrt(K)).bfloat16(), _rand((N,), 3)
    out = torch.full((M + 3, N), 7.0, dtype=torch.bfloat16, device="cuda")
    bf16s.gemm_nt(x.cuda(), w.cuda(), b.cuda(), out=out[:M])
    torch.cuda.synchronize()
    assert float((out[M:].float() - 7.0).abs().max()) == 0.0
    if M * N <= 6_000_000:
        ref = x.double() @ w.double().T + b.double()
        _close_bf16(out[:M], ref, "gemm_nt epi 0")
    again = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    bf16s.gemm_nt(x.cuda(), w.cuda(), b.cuda(), out=again)
    assert torch.equal(out[:M], again)                       # run to run reproducible


def test_gemm_nt_identity_asymmetric():
    """x = I: the result is w^T exactly (bf16 values survive the fp32 accumulate and the bf16 store)"""
    from egotap_amd import bf16s
    K = 256
    x = torch.eye(K).bfloat16()
    w = _rand((512, K), 9).bfloat16()
    out = bf16s.gemm_nt(x.cuda(), w.cuda(), None)
    assert torch.equal(out.cpu(), w.T.contiguous())


@pytest.mark.parametrize("M,N,K", [(300, 256, 64), (1153, 1024, 1024), (2304, 1024, 4096)])
def test_gemm_nt_residual_f32(M, N, K):
    """epi 1: out f32 = x w^T + b + R, in place on R (attention output / MLP down projection with the residual stream in fp32)"""
    from egotap_amd import bf16s
    x, w, b, r = _rand((M, K), 11).bfloat16(), (_rand((N, K), 12) / math.sqrt(K)).bfloat16(), _rand((N,), 13), _rand((M, N), 14, -3, 3)
    ref = x.double() @ w.double().T + b.double() + r.double()
    rc = r.cuda()
    out = bf16s.gemm_nt(x.cuda(), w.cuda(), b.cuda(), epi="residual", aux=rc, out=rc)
    assert out.data_ptr() == rc.data_ptr()
    assert float((out.double().cpu() - ref).abs().max()) < 2e-6 * math.sqrt(K) + 1e-6


@pytest.mark.parametrize("M,N,K", [(300, 256, 64), (1153, 4096, 1024)])
def test_gemm_nt_gelu_save_and_grad(M, N, K):
    """epi 2: z = x w^T + b (bf16), h = GELU(z) (bf16); epi 3: out = (dy w2^T) * GELU'(z) (the MLP's backward through the saved z)"""
    from egotap_amd import bf16s
    x, w, b = _rand((M, K), 21).bfloat16(), (_rand((N, K), 22, -2, 2) / math.sqrt(K)).bfloat16(), _rand((N,), 23)
    z, hdn = bf16s.gemm_nt(x.cuda(), w.cuda(), b.cuda(), epi="gelu_save")
    zr = x.double() @ w.double().T + b.double()
    _close_bf16(z, zr, "z")
    _close_bf16(hdn, _gelu(zr), "gelu(z)", abs_=2e-5)
    dy, w2 = _rand((M, 256), 24).bfloat16(), (_rand((N, 256), 25) / 16).bfloat16()
    dz = bf16s.gemm_nt(dy.cuda(), w2.cuda(), None, epi="gelu_grad", aux=z)
    ref = (dy.double() @ w2.double().T) * _dgelu(z.double().cpu())
    _close_bf16(dz, ref, "dz", abs_=2e-5)


@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (257, 512, 192), (1153, 1024, 1024), (2304, 1024, 4096), (5000, 3072, 1024), (70000, 1024, 256),
                                   (66000, 4096, 1024)])
def test_gemm_nt_64_deep_kernel_has_the_bits_of_the_32_deep_one(M, N, K):
    """csrc/gemm_bf16s64.h (64-deep K-tiles, 128-byte DMA row segments, quadrant phases over two K-tiles of LDS) against csrc/gemm_bf16s.h
    (32-deep ring): both run the same MFMAs in the same k order per output element, so EVERY epilogue's outputs must be bit-identical --
    two K-tiles only (fewer than the pipeline's depth), ragged M (clamped rows), one tile, tiles < CUs, many tiles per workgroup (the
    operand streams run on across tile boundaries, the store allowance after an exact epilogue), the column sums of the GELU-grad epilogue;
    canary rows catch stores past M.  Run-to-run bits of the 64-deep kernel too (a DMA / fragment-read race shows as a flicker)."""
    from egotap_amd import bf16s, lib
    L = lib.load()
    x, w, b = _rand((M, K), 41).bfloat16().cuda(), (_rand((N, K), 42, -2, 2) / math.sqrt(K)).bfloat16().cuda(), _rand((N,), 43).cuda()
    r = _rand((M, N), 44, -3, 3).cuda()
    zsave = (_rand((M, N), 45, -3, 3)).bfloat16().cuda()

    def run_all():
        o = {}
        pad = torch.full((M + 2, N), 5.0, dtype=torch.bfloat16, device="cuda")
        bf16s.gemm_nt(x, w, b, out=pad[:M])
        o["bf16"], o["canary"] = pad[:M].clone(), pad[M:].clone()
        o["res"] = bf16s.gemm_nt(x, w, b, epi="residual", aux=r, out=torch.empty_like(r))
        o["z"], o["h"] = (t.clone() for t in bf16s.gemm_nt(x, w, b, epi="gelu_save"))
        cs = torch.empty(N, device="cuda")
        o["dz"] = bf16s.gemm_nt(x, w, None, epi="gelu_grad", aux=zsave, colsum_out=cs)
        o["dz_colsum"] = cs
        o["f32"] = bf16s.gemm_nt(x, w, b, epi="f32")
        torch.cuda.synchronize()
        return o
    try:
        lib.check(L.egotap_debug_gemm_bk(32))
        old = run_all()
        lib.check(L.egotap_debug_gemm_bk(64))
        new = run_all()
        again = run_all()
    finally:
        lib.check(L.egotap_debug_gemm_bk(0))
    assert float((new["canary"].float() - 5.0).abs().max()) == 0.0
    for k in old:
        assert torch.equal(old[k], new[k]), (k, float((old[k].float() - new[k].float()).abs().max()))
        assert torch.equal(new[k], again[k]), k


def test_gemm_nt_f32_out():
    from egotap_amd import bf16s
    M, N, K = 930, 2048, 1024
    x, w, b = _rand((M, K), 31).bfloat16(), (_rand((N, K), 32) / math.sqrt(K)).bfloat16(), _rand((N,), 33)
    out = bf16s.gemm_nt(x.cuda(), w.cuda(), b.cuda(), epi="f32")
    ref = x.double() @ w.double().T + b.double()
    assert out.dtype == torch.float32 and float((out.double().cpu() - ref).abs().max()) < 2e-6 * math.sqrt(K) + 1e-6


@pytest.mark.parametrize("M,N,K", [(32, 256, 256), (100, 256, 512), (1000, 512, 256), (4096, 1024, 1024), (36864, 1024, 256), (9001, 256, 1024)])
def test_gemm_tn(M, N, K):
    """dw = dy^T x (fp32) from bf16 operands: ragged M (zero-page rows), one step, fewer steps than ring stages, many splits; an
    asymmetric pair catches swapped operands / transposed fragments; accumulate adds onto dw; run to run bit-identical"""
    from egotap_amd import bf16s
    dy, x = _rand((M, N), 41).bfloat16(), _rand((M, K), 42).bfloat16()
    ref = dy.double().T @ x.double()
    dw = torch.full((N, K), 3.0, device="cuda")
    bf16s.gemm_tn(dy.cuda(), x.cuda(), dw)
    tol = 3e-6 * math.sqrt(M) * 1.0 + 1e-5
    assert float((dw.double().cpu() - ref).abs().max()) < tol, (float((dw.double().cpu() - ref).abs().max()), tol)
    again = torch.empty_like(dw)
    bf16s.gemm_tn(dy.cuda(), x.cuda(), again)
    assert torch.equal(dw, again)
    bf16s.gemm_tn(dy.cuda(), x.cuda(), again, accumulate=True)
    assert float((again.double().cpu() - 2 * ref).abs().max()) < 2 * tol


def test_gemm_tn_strided_dy():
    """dy as a column slice of a wider matrix (the fused q|k|v gradient: three weight gradients from one [M, 3D] buffer)"""
    from egotap_amd import bf16s
    M, N, K = 2304, 256, 512
    big, x = _rand((M, 3 * N), 51).bfloat16(), _rand((M, K), 52).bfloat16()
    bc = big.cuda()
    for sidx in range(3):
        dw = torch.empty((N, K), device="cuda")
        bf16s.gemm_tn(bc[:, sidx * N:(sidx + 1) * N], x.cuda(), dw)
        ref = big[:, sidx * N:(sidx + 1) * N].double().T @ x.double()
        assert float((dw.double().cpu() - ref).abs().max()) < 3e-6 * math.sqrt(M) + 1e-5


def test_layernorm_fwd_bwd_bf16():
    """LayerNorm(1024, eps 1e-12) with a bf16 output, and its backward from a bf16 dy: dx (fp32 + bf16 copy), dgamma, dbeta and
    the column sums of dx, against float64 autograd on the same (rounded) dy"""
    from egotap_amd import bf16s
    rows, D = 777, 1024
    x, g, b = _rand((rows, D), 61, -3, 3), _rand((D,), 62, 0.5, 1.5), _rand((D,), 63)
    dy, dres = _rand((rows, D), 64).bfloat16(), _rand((rows, D), 65)
    y, mean, rstd = bf16s.layernorm_fwd(x.cuda(), g.cuda(), b.cuda())
    xd = x.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xd, (D,), gd, bd, 1e-12)
    _close_bf16(y, yr.detach(), "layernorm y")
    yr.backward(dy.double())
    dgam, dbet, dcs = (torch.empty(D, device="cuda") for _ in range(3))
    dx, dxb = bf16s.layernorm_bwd(x.cuda(), dy.cuda(), g.cuda(), mean, rstd, dgam, dbet, dres=dres.cuda(), dcolsum=dcs)
    ref_dx = xd.grad + dres.double()
    assert float((dx.double().cpu() - ref_dx).abs().max()) < 2e-5
    _close_bf16(dxb, ref_dx, "dx bf16")
    assert float((dgam.double().cpu() - gd.grad).abs().max()) < 2e-4 and float((dbet.double().cpu() - bd.grad).abs().max()) < 2e-4
    assert float((dcs.double().cpu() - ref_dx.sum(0)).abs().max()) < 2e-4


def test_colsum_and_prep_weight():
    from egotap_amd import bf16s
    y = _rand((5003, 3072), 71).bfloat16()
    out = torch.zeros(1024, device="cuda")
    bf16s.colsum(y.cuda()[:, 1024:2048], out)                         # a column slice (fused q|k|v gradient)
    assert float((out.double().cpu() - y[:, 1024:2048].double().sum(0)).abs().max()) < 2e-3
    w = _rand((1000, 520), 72)
    wb, wt = torch.empty((1000, 520), dtype=torch.bfloat16, device="cuda"), torch.empty((520, 1000), dtype=torch.bfloat16, device="cuda")
    bf16s.prep_weight(w.cuda(), wb, wt)
    assert torch.equal(wb.cpu(), w.bfloat16()) and torch.equal(wt.cpu(), w.bfloat16().T.contiguous())
    assert torch.equal(bf16s.from_f32(w.cuda()).cpu(), w.bfloat16())


@pytest.mark.parametrize("B,N", [(2, 576), (1, 2304), (3, 96), (5, 96), (1, 32)])
def test_attention_fwd_bwd_bf16(B, N):
    """softmax attention on bf16 q|k|v: ctx, log-sum-exp and the three gradients (two backward kernels) against float64 autograd
    on the same rounded inputs; a structured V (value = key index) catches permuted keys"""
    from egotap_amd import bf16s
    heads, dh = 8, 128
    D = heads * dh
    qkv = _rand((B * N, 3 * D), 81, -2, 2).bfloat16()
    qkv[:, 2 * D:2 * D + 4] = (torch.arange(B * N) % N).to(torch.bfloat16)[:, None] / 64.0
    dctx = _rand((B * N, D), 82).bfloat16()
    ctx, lse = bf16s.attention_fwd(qkv.cuda(), B, N, heads)
    t = qkv.double().requires_grad_(True)
    q, k, v = (t[:, i * D:(i + 1) * D].view(B, N, heads, dh).transpose(1, 2) for i in range(3))
    sc = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B * N, D)
    # probabilities and V are bf16 operands (2^-9 each) of a 576-term sum with |v| <= 2: absolute noise of a few 1e-3
    _close_bf16(ctx, ref.detach(), "ctx", abs_=4e-3)
    assert float((lse.double().cpu() - torch.logsumexp(sc, -1).reshape(-1).detach()).abs().max()) < 2e-3
    # backward from the bf16 ctx the forward stored (what the training step does)
    ref.backward(dctx.double())
    dqkv = bf16s.attention_bwd(qkv.cuda(), ctx, dctx.cuda(), lse, B, N, heads)
    err = (dqkv.double().cpu() - t.grad).abs()
    scale = t.grad.abs().mean()
    assert float(err.max()) < 0.1 * float(t.grad.abs().max()) and float(err.mean()) < 0.02 * float(scale), (float(err.max()), float(err.mean()), float(scale))
    for i, name in enumerate(("dq", "dk", "dv")):
        a, r = dqkv[:, i * D:(i + 1) * D].double().cpu().reshape(-1), t.grad[:, i * D:(i + 1) * D].reshape(-1)
        cos = float(a @ r / (a.norm() * r.norm()))
        assert cos > 0.999, (name, cos)
    again = bf16s.attention_bwd(qkv.cuda(), ctx, dctx.cuda(), lse, B, N, heads)
    assert torch.equal(dqkv, again)


@pytest.mark.parametrize("B,N", [(2, 576), (1, 96)])
def test_attention_forward_running_maximum_rescale_with_a_pending_tile(B, N):
    """The forward defers P V of a key tile by one step and runs it under the next tile's exponentials (attention_bf16s2.h); the running
    maximum is raised only when a tile exceeds it by 2^6.  On bounded random scores that branch fires at the first tile only, so a passing
    random test says nothing about it (cdna_hip_programming.md T13: nothing pending may be left at the old scale).  Here the scores of a row
    GROW along the keys by `slope` per key in log2 units -- a rescale at every tile with the previous tile's P V still pending (slope 0.5, 2),
    at every few tiles (0.05), never after the first (negative slopes), all mixed over the 32 rows of one wave's query block, heads with
    noise on top; context and log-sum-exp against float64 softmax on the same bf16 operands."""
    from egotap_amd import bf16s
    heads, dh = 8, 128
    D = heads * dh
    g = torch.Generator().manual_seed(91)
    u = torch.randn(dh, generator=g)
    u = u / u.norm()
    slopes = torch.tensor([0.5, -0.5, 2.0, 0.05, -0.05, 0.0, 1.0, -2.0])[torch.arange(B * N) % 8] * (1.0 + (torch.arange(B * N) % 5).float() / 4.0)
    ln2 = math.log(2.0)
    qkv = torch.randn(B * N, 3 * D, generator=g) * 0.05
    key_pos = (torch.arange(B * N) % N).float()
    for h in range(heads):
        qkv[:, h * dh:(h + 1) * dh] += (slopes * ln2 * math.sqrt(dh))[:, None] * u[None, :] * (1.0 if h % 2 == 0 else 0.5)      # q_i = slope_i sqrt(dh) ln2 u
        qkv[:, D + h * dh:D + (h + 1) * dh] += key_pos[:, None] * u[None, :]                                                    # k_j = j u  ->  q.k / sqrt(dh) = slope_i j ln2
    qkv[:, 2 * D:] = torch.randn(B * N, D, generator=g)
    qkv[:, 2 * D:2 * D + 4] = (key_pos / 64.0)[:, None]
    qkv = qkv.bfloat16()
    ctx, lse = bf16s.attention_fwd(qkv.cuda(), B, N, heads)
    again, lse2 = bf16s.attention_fwd(qkv.cuda(), B, N, heads)
    assert torch.equal(ctx, again) and torch.equal(lse, lse2)
    t = qkv.double()
    q, k, v = (t[:, i * D:(i + 1) * D].view(B, N, heads, dh).transpose(1, 2) for i in range(3))
    sc = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    growth = float((sc[0, 0, 2, 1:] - sc[0, 0, 2, :-1]).mean() / ln2)               # row 2: slope 2 x gain: the maximum rises by > 2^6 every tile
    assert growth * 32 > 6.0, growth
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B * N, D)
    err = (ctx.double().cpu() - ref).abs()
    # one bf16 rounding of the stored context (|v| up to ~4; up to 9 in the key-index columns) + bf16 P and V in a peaked sum; a tile left at the
    # wrong scale is an O(0.1 ... 1) error on every row whose maximum moved
    excess = float((err - (2.0 ** -7 * ref.abs() + 1.5e-2)).max())
    assert excess <= 0.0 and float(err.mean()) < 2e-3, (excess, float(err.max()), float(err.mean()))
    ref_lse = torch.logsumexp(sc, -1).reshape(-1)
    assert float((lse.double().cpu() - ref_lse).abs().max()) < 2e-3 * max(1.0, float(ref_lse.abs().max()) / 64.0)


@pytest.mark.parametrize("B,N", [(3, 576), (2, 96)])
def test_attention_backward_bias_sums_equal_the_column_sums_of_dqkv(B, N):
    """egotap_bf16_attention_bwd_bias: the q | k | v bias gradients.  N % 64 == 0: per-block partial sums written by the dQ / dK+dV
    kernels' epilogues; other multiples of 32 (N = 96): the column-sum pass over dqkv.  Both against the float64 column sums of the
    dqkv tensor the call returns, which must have the bits of the plain backward."""
    from egotap_amd import bf16s
    heads, D = 8, 1024
    torch.manual_seed(5)
    qkv = (torch.randn(B * N, 3 * D, device="cuda") * 0.5).bfloat16()
    dctx = (torch.randn(B * N, D, device="cuda") * 0.1).bfloat16()
    ctx, lse = bf16s.attention_fwd(qkv, B, N, heads)
    gb = tuple(torch.full((D,), float("nan"), device="cuda") for _ in range(3))
    dqkv = bf16s.attention_bwd(qkv, ctx, dctx, lse, B, N, heads, bias_grads=gb)
    plain = bf16s.attention_bwd(qkv, ctx, dctx, lse, B, N, heads)
    torch.cuda.synchronize()
    assert torch.equal(dqkv, plain)
    ref, mag = dqkv.double().sum(0), dqkv.double().abs().sum(0)
    for q in range(3):
        got, want = gb[q].double(), ref[q * D:(q + 1) * D]
        assert float((got - want).abs().max()) <= 2e-6 * float(mag[q * D:(q + 1) * D].max()) + 1e-9, q


@pytest.mark.parametrize("preset,hm,B", [("UnrealEgo", 64, 3), ("EgoCap", 128, 1)])
def test_patch_embedding_on_the_bf16_storage_gemm(preset, hm, B):
    """egotap_bf16_patch_fwd: ViTPatchEmbeddings + mask token + position embeddings over the tiled heatmap image (net_architecture.py:326-336,
    modeling_vit.py:137-153) with the bf16 heatmaps fetched patch row by patch row through the LDS DMA (XPatch) -- against float64
    arithmetic on the same bf16-rounded heatmaps and weights (oracle.vit_embed's tiling): real cells, dummy cells (mask token), every
    token of every frame; bit-reproducible."""
    import ctypes as C
    from egotap_amd import bf16s, lib, spec
    from egotap_amd.synthetic import synth_input
    from gpu_util import lift_net
    from oracle import lift_ref as O
    L = lib.load()
    net, sd_np, p = lift_net(preset, hm)
    h = net._ensure_handle()
    D, seq, J = p.vit_dim, p.seq, p.n_joints_hm
    hm_t = torch.from_numpy(synth_input(f"hm_patch_{hm}", (B, p.in_channels, hm, hm)))
    v = "pos_heatmap_encoder.vit.embeddings."
    w = torch.from_numpy(sd_np[v + "patch_embeddings.projection.weight"]).reshape(D, 256)
    b, mt, pos = (torch.from_numpy(sd_np[v + k]).reshape(-1) for k in ("patch_embeddings.projection.bias", "mask_token", "position_embeddings"))
    hm_dev = hm_t.cuda()
    hmb = bf16s.from_f32(hm_dev)
    wb = torch.empty((D, 256), dtype=torch.bfloat16, device="cuda")
    wc, bc, mc, pc = w.cuda(), b.cuda(), mt.cuda(), pos.cuda()          # (kept alive: the call takes raw pointers)
    bf16s.prep_weight(wc, wb)
    torch.cuda.synchronize()
    zp = bf16s.zero_page(torch.device("cuda", torch.cuda.current_device()))
    outs = []
    for _ in range(2):
        x = torch.full((B * seq, D), float("nan"), device="cuda")
        lib.check(L.egotap_bf16_patch_fwd(h, C.c_void_p(hmb.data_ptr()), C.c_void_p(wb.data_ptr()), C.c_void_p(bc.data_ptr()), C.c_void_p(mc.data_ptr()),
                                          C.c_void_p(pc.data_ptr()), C.c_void_p(zp.data_ptr()), C.c_void_p(x.data_ptr()), B,
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        outs.append(x.clone())
    assert torch.equal(outs[0], outs[1])
    rb = lambda t: t.float().bfloat16().double()
    patches, dummy = O.tile_to_patches(rb(hm_t[:, : 2 * J]), p)
    emb = patches @ rb(w).T + b.double()
    emb = torch.where(dummy.view(1, -1, 1), mt.double().view(1, 1, D), emb) + pos.double().view(1, -1, D)[:, -seq:]
    got = outs[0].double().cpu().view(B, seq, D)
    err = float((got - emb).abs().max())
    assert err < 2e-5 * float(emb.abs().max()) + 1e-6, err


@pytest.mark.parametrize("B", [16, 3])
def test_fc1_weight_gradients_on_gathered_rows(B):
    """egotap_bf16_fc1_wgrad (weight gradient of fc1 of both encoders: dW[n, k] = sum over rows of dz[row, n] X[row, k], X gathered by the
    TXTokens / TXRot loaders of gemm_tn_bf16s.h -- net_architecture.py:388-406, 690-694) against float64 on the same bf16 operands.
    B = 16: 480 rows = whole 32-row steps -> [r5] the scalar-base form of the gather (wave-uniform tensor base + one 32-bit lane offset);
    B = 3: 90 rows, ragged last step -> the general form (per-lane pointers, zero page)."""
    from egotap_amd import bf16s
    from gpu_util import lift_net
    net, _, p = lift_net("UnrealEgo")
    h = net._ensure_handle()
    T, D, seq, side, ppd, grid, J = p.tokens, p.vit_dim, p.seq, p.side, p.ppd, p.grid, p.n_joints_hm
    HW = p.hm_size * p.hm_size
    g = torch.Generator().manual_seed(17)
    tokens = (torch.randn(B * seq, D, generator=g) * 0.5).bfloat16()
    hmb = (torch.rand(B, p.in_channels, HW, generator=g)).bfloat16()
    dz = (torch.randn(B * T, 2048, generator=g) * 0.1).bfloat16()
    # position encoder: row (b, i), k = (patch s of heatmap i, channel c)
    tok = tokens.double().view(B, side, side, D)
    rows = []
    for i in range(T):
        r0, c0 = ppd * (i // grid), ppd * (i % grid)
        rows.append(tok[:, r0:r0 + ppd, c0:c0 + ppd, :].reshape(B, ppd * ppd * D))
    Xp = torch.stack(rows, 1).reshape(B * T, ppd * ppd * D)
    # limb encoder: row (b, eye J + j), k = (cos | sin, pixel)
    hmd = hmb.double()
    rows = []
    for eye in range(2):
        for j in range(J):
            rows.append(torch.cat([hmd[:, 2 * J + eye * 2 * J + j], hmd[:, 2 * J + eye * 2 * J + J + j]], 1))
    Xr = torch.stack(rows, 1).reshape(B * T, 2 * HW)
    for which, src, X in ((0, tokens, Xp), (1, hmb, Xr)):
        K = X.shape[1]
        dw = torch.full((2048, K), float("nan"), device="cuda")
        bf16s.fc1_wgrad(h, which, dz.cuda(), src.cuda(), dw, B)
        again = torch.full((2048, K), float("nan"), device="cuda")
        bf16s.fc1_wgrad(h, which, dz.cuda(), src.cuda(), again, B)
        torch.cuda.synchronize()
        assert torch.equal(dw, again)
        ref = dz.double().T @ X
        err = float((dw.double().cpu() - ref).abs().max())
        assert err < 2e-5 * float(ref.abs().max()) + 1e-5, (which, B, err, float(ref.abs().max()))


@pytest.mark.parametrize("B", [3, 20])
def test_fc1_forward_on_the_64_deep_kernel_has_the_bits_of_the_32_deep_one(B):
    """[r5] egotap_bf16_fc1_fwd (fc1 of both encoders: rows gathered from the ViT tokens / the limb heatmaps) runs on the 64-deep GEMM through the
    X64Tokens / X64Rot loaders (gemm_bf16s64.h); pinned to the 32-deep kernel (egotap_debug_gemm_bk(32), XTokens / XRot) it must give the SAME bits --
    one k order per output element -- and both must equal float64 on the same bf16 operands.  B = 3: a ragged 90-row tile; B = 20: 600 rows."""
    from egotap_amd import bf16s, lib
    from gpu_util import lift_net
    L = lib.load()
    net, _, p = lift_net("UnrealEgo")
    h = net._ensure_handle()
    T, D, seq, side, ppd, grid, J = p.tokens, p.vit_dim, p.seq, p.side, p.ppd, p.grid, p.n_joints_hm
    HW = p.hm_size * p.hm_size
    g = torch.Generator().manual_seed(23)
    tokens = (torch.randn(B * seq, D, generator=g) * 0.5).bfloat16()
    hmb = torch.rand(B, p.in_channels, HW, generator=g).bfloat16()
    bias = torch.randn(2048, generator=g)
    tok = tokens.double().view(B, side, side, D)
    Xp = torch.stack([tok[:, ppd * (i // grid):ppd * (i // grid) + ppd, ppd * (i % grid):ppd * (i % grid) + ppd, :].reshape(B, ppd * ppd * D) for i in range(T)], 1)
    hmd = hmb.double()
    Xr = torch.stack([torch.cat([hmd[:, 2 * J + e * 2 * J + j], hmd[:, 2 * J + e * 2 * J + J + j]], 1) for e in range(2) for j in range(J)], 1)
    for which, src, X in ((0, tokens, Xp.reshape(B * T, -1)), (1, hmb, Xr.reshape(B * T, -1))):
        K = X.shape[1]
        w = (torch.randn(2048, K, generator=g) * 0.02).bfloat16()
        outs = {}
        try:
            for bk in (32, 0):
                lib.check(L.egotap_debug_gemm_bk(bk))
                outs[bk] = bf16s.fc1_fwd(h, which, src.cuda(), w.cuda(), bias.cuda(), B, T).clone()
            lib.check(L.egotap_debug_conv_addressing(1))      # [r5] the 64-deep kernel's gather by per-lane pointers (X64Tokens / X64Rot) instead of a scalar origin
            outs["ptr"] = bf16s.fc1_fwd(h, which, src.cuda(), w.cuda(), bias.cuda(), B, T).clone()
        finally:
            lib.check(L.egotap_debug_conv_addressing(0))
            lib.check(L.egotap_debug_gemm_bk(0))
        torch.cuda.synchronize()
        assert torch.equal(outs[32], outs[0]) and torch.equal(outs["ptr"], outs[0]), which
        ref = X @ w.double().T + bias.double()
        err = float((outs[0].double().cpu() - ref).abs().max())
        assert err < 2e-5 * float(ref.abs().max()) + 1e-5, (which, B, err)
