"""GPU parity of the single HIP operators (through the C ABI) against float64 CPU math."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g, dtype=torch.float64) * (hi - lo) + lo).float()


def _close(got, ref64, atol, rtol=1e-5):
    got = got.detach().cpu().double()
    err = (got - ref64).abs()
    tol = atol + rtol * ref64.abs()
    assert bool((err <= tol).all()), f"max err {err.max().item():.3e} (max ref {ref64.abs().max().item():.3e})"


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (576, 1024, 256), (1, 128, 64), (130, 256, 96), (1152, 3072, 1024),
                                    (77, 512, 2048)])
def test_linear_bias(M, N, K):
    from egotap_amd import lib
    x, w, b = _rand((M, K), 1), _rand((N, K), 2, -0.1, 0.1), _rand((N,), 3)
    y = lib.linear(x.cuda(), w.cuda(), b.cuda())
    ref = x.double() @ w.double().T + b.double()
    _close(y, ref, atol=2e-6 * math.sqrt(K))


def test_linear_identity_asymmetric():
    """A = I against an asymmetric W catches a transposed C write (cdna_hip_programming.md section 3)."""
    from egotap_amd import lib
    K = 128
    x = torch.eye(K)
    w = torch.arange(256 * K, dtype=torch.float32).reshape(256, K) * 0.5
    y = lib.linear(x.cuda(), w.cuda(), torch.zeros(256).cuda()).cpu()
    assert torch.equal(y, w.T.contiguous())


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 6, 7, 8, 9, 10, 11, 12])
def test_linear_tiles_agree(tile):
    from egotap_amd import lib
    M, N, K = 700, 512, 160
    x, w, b = _rand((M, K), 11), _rand((N, K), 12, -0.1, 0.1), _rand((N,), 13)
    ref = x.double() @ w.double().T + b.double()
    y = lib.linear(x.cuda(), w.cuda(), b.cuda(), tile=tile)
    _close(y, ref, atol=3e-5)


def test_linear_epilogues():
    from egotap_amd import lib
    M, N, K = 200, 256, 512
    x, w, b = _rand((M, K), 21), _rand((N, K), 22, -0.05, 0.05), _rand((N,), 23)
    z = x.double() @ w.double().T + b.double()
    r = _rand((M, N), 24)
    _close(lib.linear(x.cuda(), w.cuda(), b.cuda(), "residual", residual=r.cuda()), z + r.double(), 3e-5)
    gel = 0.5 * z * (1.0 + torch.erf(z / math.sqrt(2.0)))
    _close(lib.linear(x.cuda(), w.cuda(), b.cuda(), "gelu"), gel, 3e-5)
    g, beta, mean, var = _rand((N,), 25, 0.5, 1.5), _rand((N,), 26), _rand((N,), 27, -0.2, 0.2), _rand((N,), 28, 0.5, 1.5)
    bn = (z - mean.double()) / torch.sqrt(var.double() + 1e-5) * g.double() + beta.double()
    bn = torch.where(bn > 0, bn, 0.2 * bn)
    _close(lib.linear(x.cuda(), w.cuda(), b.cuda(), "bn_lrelu", bn=tuple(t.cuda() for t in (g, beta, mean, var))), bn, 5e-5)
    # residual in place (the transformer's x += ...): output buffer aliases the residual
    xr = r.cuda().clone()
    import ctypes as C
    L = lib.load()
    xc, wc, bc = x.cuda(), w.cuda(), b.cuda()
    lib.check(L.egotap_linear_f32(C.c_void_p(xc.data_ptr()), C.c_void_p(wc.data_ptr()), C.c_void_p(bc.data_ptr()),
                                  C.c_void_p(xr.data_ptr()), M, N, K, 1, C.c_void_p(xr.data_ptr()), None, None, None, None, 0,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    _close(xr, z + r.double(), 3e-5)


def test_linear_rejects_bad_shapes():
    from egotap_amd import lib
    x, w, b = torch.zeros(4, 40).cuda(), torch.zeros(128, 40).cuda(), torch.zeros(128).cuda()
    with pytest.raises(lib.EgotapError):
        lib.linear(x, w, b)                      # K = 40 not a multiple of 32
    with pytest.raises(lib.EgotapError):
        lib.linear(torch.zeros(4, 64), w, b)     # CPU tensor


@pytest.mark.parametrize("rows", [1, 5, 576, 4099])
def test_layernorm(rows):
    from egotap_amd import lib
    x, g, b = _rand((rows, 1024), 31, -3, 3), _rand((1024,), 32, 0.5, 1.5), _rand((1024,), 33)
    xd = x.double()
    mu = xd.mean(-1, keepdim=True)
    var = ((xd - mu) ** 2).mean(-1, keepdim=True)
    ref = (xd - mu) / torch.sqrt(var + 1e-12) * g.double() + b.double()
    _close(lib.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-12), ref, 3e-6)


@pytest.mark.parametrize("B,N,heads", [(1, 32, 1), (2, 576, 8), (1, 64, 2), (3, 96, 8)])
def test_attention(B, N, heads):
    from egotap_amd import lib
    D = heads * 128
    qkv = _rand((B * N, 3 * D), 41, -2, 2)
    q, k, v = [t.reshape(B, N, heads, 128).transpose(1, 2).double() for t in qkv.split(D, dim=1)]
    s = q @ k.transpose(-1, -2) / math.sqrt(128.0)
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * N, D)
    _close(lib.attention(qkv.cuda(), B, N, heads), ref, 3e-6)


def test_attention_rescale_branch():
    """Force the running max to jump at a late key tile (online-softmax rescale path)."""
    from egotap_amd import lib
    B, N, heads, D = 1, 128, 1, 128
    qkv = _rand((N, 3 * D), 51, -0.5, 0.5)
    qkv[7, :128] = 3.0                    # query 7 ...
    qkv[100, 128:256] = 3.0               # ... strongly matches key 100 (4th key tile)
    q, k, v = [t.reshape(B, N, heads, 128).transpose(1, 2).double() for t in qkv.split(D, dim=1)]
    s = q @ k.transpose(-1, -2) / math.sqrt(128.0)
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(N, D)
    _close(lib.attention(qkv.cuda(), B, N, heads), ref, 3e-6)


@pytest.mark.parametrize("M,N,K", [(256, 256, 16), (300, 512, 48), (1153, 768, 1024), (5000, 256, 32), (2304, 1024, 64)])
def test_linear_persistent_tile_many_shapes(M, N, K):
    """persistent 256x256 kernel: tiles < CUs, ragged M, short K (fewer slabs than pipeline stages), many tiles per block"""
    from egotap_amd import lib
    x, w, b = _rand((M, K), 61), _rand((N, K), 62, -0.1, 0.1), _rand((N,), 63)
    ref = x.double() @ w.double().T + b.double()
    y = lib.linear(x.cuda(), w.cuda(), b.cuda(), tile=12)
    _close(y, ref, atol=2e-6 * math.sqrt(K) + 1e-6)
    y1 = lib.linear(x.cuda(), w.cuda(), b.cuda(), tile=8)
    assert torch.equal(y, y1)            # same k order as the 128x128 pipelined kernel: bit-identical


@pytest.mark.parametrize("M,N,K", [(256, 256, 16), (300, 512, 48), (1153, 768, 1024), (5000, 256, 32), (2304, 1024, 64), (70000, 1024, 256)])
def test_linear_f32_dma_bit_identical(M, N, K):
    """gemm_f32_dma_kernel (tile 19: slabs staged global -> LDS by DMA, xor-swizzled unpadded rows, four stages, barrier mid-slab):
    the k order per output element is that of the register-staged kernels, so the result is bit for bit theirs -- fewer slabs than
    stages, a single slab per tile, ragged M, tiles < CUs, several tiles per workgroup."""
    from egotap_amd import lib
    x, w, b = _rand((M, K), 71), _rand((N, K), 72, -0.1, 0.1), _rand((N,), 73)
    y = lib.linear(x.cuda(), w.cuda(), b.cuda(), tile=19)
    assert torch.equal(y, lib.linear(x.cuda(), w.cuda(), b.cuda(), tile=12))
    if M * N <= 4_000_000:
        _close(y, x.double() @ w.double().T + b.double(), atol=2e-6 * math.sqrt(K) + 1e-6)


def test_linear_dma_kernels_random_shapes():
    """seeded sweep over odd row counts and every slab count around the pipeline depth: both DMA-staged GEMMs reproduce their
    register-staged counterparts bit for bit (fp32: tile 19 vs 12; bf16: egotap_linear_bf16_dma on bf16 copies vs tile 16)"""
    from egotap_amd import lib
    rng = np.random.default_rng(20240607)
    for trial in range(24):
        M = int(rng.integers(1, 3000))
        N = int(rng.choice([256, 512, 768, 1024]))
        K32 = int(rng.integers(1, 12)) * 32
        x, w, b = _rand((M, K32), 200 + trial), _rand((N, K32), 300 + trial, -0.2, 0.2), _rand((N,), 400 + trial)
        xc, wc, bc = x.cuda(), w.cuda(), b.cuda()
        assert torch.equal(lib.linear(xc, wc, bc, tile=19), lib.linear(xc, wc, bc, tile=12)), (M, N, K32)
        assert torch.equal(lib.linear_bf16_dma(xc.bfloat16(), wc.bfloat16(), bc), lib.linear(xc, wc, bc, tile=16)), (M, N, K32)
        K16 = K32 + 16                                   # fp32 slabs are 16 deep: odd slab counts too
        x, w = _rand((M, K16), 500 + trial), _rand((N, K16), 600 + trial, -0.2, 0.2)
        assert torch.equal(lib.linear(x.cuda(), w.cuda(), bc, tile=19), lib.linear(x.cuda(), w.cuda(), bc, tile=12)), (M, N, K16)


# ---- bf16 matrix-core GEMMs with fp32 operands in HBM (gemm_bf16.h): tile 13 = bf16x3 split, tile 14 = plain bf16
@pytest.mark.parametrize("M,N,K", [(1000, 512, 1024), (300, 256, 16), (257, 256, 48), (2048, 1024, 4096)])
def test_linear_bf16x3_error_model(M, N, K):
    """hi+lo split keeps 16 significant bits per operand and drops a_lo*b_lo: error <= ~2^-16 per product, random signs.
    Checked against float64: rms error within 8x of the exact-fp32 kernel's and far below plain bf16's."""
    from egotap_amd import lib
    x, w, b = _rand((M, K), 31), _rand((N, K), 32, -1.0, 1.0) / math.sqrt(K), _rand((N,), 33)
    ref = x.double() @ w.double().T + b.double()
    scale = float((x.double().abs() @ w.double().abs().T).mean())       # sum |a||b| per output
    y3 = lib.linear(x.cuda(), w.cuda(), b.cuda(), tile=13).cpu().double()
    err3 = (y3 - ref).abs()
    assert float(err3.max()) < 2.0 ** -15 * scale, (float(err3.max()), scale)
    assert float(err3.pow(2).mean().sqrt()) < 2.0 ** -17 * scale


@pytest.mark.parametrize("M,N,K", [(1000, 512, 1024), (300, 256, 32), (257, 256, 96)])
def test_linear_bf16_plain(M, N, K):
    """plain bf16 operands (RNE), fp32 accumulate: equals the float64 product of the bf16-rounded operands to fp32 rounding"""
    from egotap_amd import lib
    x, w, b = _rand((M, K), 41), _rand((N, K), 42, -1.0, 1.0) / math.sqrt(K), _rand((N,), 43)
    xr, wr = x.bfloat16().double(), w.bfloat16().double()
    ref = xr @ wr.T + b.double()
    y = lib.linear(x.cuda(), w.cuda(), b.cuda(), tile=14)
    _close(y, ref, atol=2e-6 * math.sqrt(K))


@pytest.mark.parametrize("M,N,K", [(256, 256, 32), (300, 256, 96), (257, 512, 128), (1153, 768, 1024), (5000, 256, 64), (70000, 1024, 256)])
def test_linear_bf16_dma(M, N, K):
    """gemm_bf16_dma_kernel (egotap_linear_bf16_dma: caller-owned bf16 copies of both operands, global -> LDS DMA, xor-swizzled
    unpadded rows, 4 stages): equals the float64 product of the bf16-rounded operands to fp32 rounding -- fewer slabs than stages,
    ragged M, tiles < CUs, several tiles per workgroup (the last shape: 1100 tiles on 256 CUs) -- and bit for bit the
    register-staged bf16 kernel (tile 16: same operand rounding (RNE, as torch's .bfloat16()), same k order per output element)."""
    from egotap_amd import lib
    x, w, b = _rand((M, K), 51), _rand((N, K), 52, -1.0, 1.0) / math.sqrt(K), _rand((N,), 53)
    xb, wb = x.cuda().bfloat16(), w.cuda().bfloat16()
    y = lib.linear_bf16_dma(xb, wb, b.cuda())
    if M * N <= 4_000_000:
        ref = x.bfloat16().double() @ w.bfloat16().double().T + b.double()
        _close(y, ref, atol=2e-6 * math.sqrt(K))
    y16 = lib.linear(x.cuda(), w.cuda(), b.cuda(), tile=16)
    assert torch.equal(y, y16)
    assert torch.equal(y, lib.linear_bf16_dma(xb, wb, b.cuda()))       # run to run reproducible


def test_linear_bf16x3_identity_asymmetric_and_ragged_rows():
    """A = I: hi+lo of 1.0 is exact, so the result is W^T rounded to 16 bits of mantissa; rows past M are never written."""
    from egotap_amd import lib
    K = 128
    x = torch.eye(K)
    w = (torch.arange(256 * K, dtype=torch.float32).reshape(256, K) * 0.5)
    y = lib.linear(x.cuda(), w.cuda(), torch.zeros(256).cuda(), tile=13).cpu()
    assert float((y - w.T).abs().max()) <= float(w.abs().max()) * 2.0 ** -16
    M, N = 300, 512
    xb = _rand((M, K), 51).cuda()
    wb, bb = _rand((N, K), 52).cuda(), _rand((N,), 53).cuda()
    full = lib.linear(xb, wb, bb, tile=13)
    part = lib.linear(xb[:37].contiguous(), wb, bb, tile=13)
    assert torch.equal(full[:37], part)          # per-row result independent of M (fixed k order)


# ---- fused evaluation metrics (metrics.h): MPJPE + Procrustes-aligned MPJPE per sample
@pytest.mark.parametrize("B,J", [(6, 16), (37, 17), (1, 16), (300, 16)])
def test_pose_metrics_matches_reference_golden_and_oracle(B, J):
    import os
    from egotap_amd import lib
    from egotap_amd.synthetic import synth_input
    from oracle import lift_ref as O
    s1 = torch.from_numpy(synth_input("procrustes_s1", (B, J, 3), -30.0, 30.0))
    s2 = torch.from_numpy(synth_input("procrustes_s2", (B, J, 3), -30.0, 30.0))
    s2[:3] = s1[:3] * 1.7 + 0.3 * s2[:3]          # as tools/make_golden.py gen_procrustes: partly correlated pairs
    e, pa, al = lib.pose_metrics(s1.cuda(), s2.cuda(), want_aligned=True)
    ref_al = O.procrustes_align(s1.double(), s2.double())
    _close(al, ref_al, atol=2e-4, rtol=1e-5)
    _close(e, torch.linalg.norm(s2.double() - s1.double(), dim=-1).mean(-1), atol=1e-4)
    _close(pa, torch.linalg.norm(s2.double() - ref_al, dim=-1).mean(-1), atol=1e-4)
    if (B, J) == (6, 16):        # the reference's own batch_compute_similarity_transform_torch on these inputs
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", "procrustes.npz"))
        np.testing.assert_allclose(al.cpu().numpy(), g["s1_hat"], atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("B,J", [(2, 16), (3, 16), (2, 17), (3, 17)])
def test_pose_metrics_reference_batch_axes_for_batches_of_two_and_three(B, J):
    """utils/util.py:337: a batch of 2 or 3 frames is aligned over the wrong axes by the reference; egotap_pose_metrics_batch_axes
    reproduces the reference's own outputs (tests/golden/procrustes_batch_axes.npz) through a 3 x 3 / 2 x 2 reduction of its J x J SVD"""
    import os
    from egotap_amd import lib
    from egotap_amd.synthetic import synth_input
    from oracle import lift_ref as O
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "procrustes_batch_axes.npz"))
    a = torch.from_numpy(synth_input(f"procrustes_q1_{B}_{J}", (B, J, 3), -30.0, 30.0))
    b = torch.from_numpy(synth_input(f"procrustes_q2_{B}_{J}", (B, J, 3), -30.0, 30.0))
    b[:1] = a[:1] * 1.3 + 0.5 * b[:1]
    e, pa, al = lib.pose_metrics(a.cuda(), b.cuda(), want_aligned=True, reference_batch_axes=True)
    np.testing.assert_allclose(al.cpu().numpy(), g[f"s1_hat_b{B}_j{J}"], atol=2e-4, rtol=1e-4)           # the reference itself (fp32 LAPACK)
    ref = O.procrustes_align_batch_axes(a.double(), b.double())
    _close(al, ref, atol=2e-5, rtol=1e-6)                                                                # the float64 restatement
    _close(pa, torch.linalg.norm(b.double() - ref, dim=-1).mean(-1), atol=1e-5)
    _close(e, torch.linalg.norm(b.double() - a.double(), dim=-1).mean(-1), atol=1e-5)
    # the switch off, or any other batch size: the 3 x 3 alignment
    e2, pa2 = lib.pose_metrics(a.cuda(), b.cuda())
    _close(pa2, torch.linalg.norm(b.double() - O.procrustes_align(a.double(), b.double()), dim=-1).mean(-1), atol=1e-4)
    a4, b4 = torch.cat([a, a])[:4], torch.cat([b, b])[:4]                  # the same frames inside a batch of 4
    _, pa4 = lib.pose_metrics(a4.cuda(), b4.cuda(), reference_batch_axes=True)
    _close(pa4[:B], pa2.double().cpu(), atol=1e-6)


def test_pose_metrics_reflection_and_identity():
    """gt = mirrored pred needs the det-sign fix (a reflection is not allowed); gt = similarity(pred) aligns to zero error"""
    from egotap_amd import lib
    from oracle import lift_ref as O
    s1 = _rand((8, 16, 3), 81, -20, 20)
    mirrored = s1.clone(); mirrored[..., 0] *= -1
    e, pa, al = lib.pose_metrics(s1.cuda(), mirrored.cuda(), want_aligned=True)
    _close(al, O.procrustes_align(s1.double(), mirrored.double()), atol=5e-4)
    assert float(pa.min()) > 0.5                       # cannot be aligned by a proper rotation
    c, s_ = math.cos(0.7), math.sin(0.7)
    R = torch.tensor([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]])
    moved = 1.7 * s1 @ R.T + torch.tensor([3.0, -2.0, 5.0])
    e, pa = lib.pose_metrics(s1.cuda(), moved.cuda())
    assert float(pa.max()) < 1e-4 and float(e.min()) > 1.0
    with pytest.raises(ValueError):
        lib.pose_metrics(s1.cuda(), moved[:, :15].cuda())


# ---- attention on the bf16 matrix cores (attention_bf16.h): transposed V reads, probability accumulators as operands
def _attn_ref(qkv, B, N, heads, rounded=False):
    D = heads * 128
    src = qkv.bfloat16().float() if rounded else qkv
    q, k, v = [t.reshape(B, N, heads, 128).transpose(1, 2).double() for t in src.split(D, dim=1)]
    s = q @ k.transpose(-1, -2) / math.sqrt(128.0)
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * N, D)


@pytest.mark.parametrize("B,N,heads", [(1, 32, 1), (2, 576, 8), (1, 64, 2), (3, 96, 8), (1, 2304, 2)])
def test_attention_bf16x3(B, N, heads):
    """hi+lo split of Q, K, V and of the probabilities: asymmetric random data (a transposed or permuted V fragment, a
    wrong key order of the probability operand or a bad lane map all show up as O(1) errors), partial last query group."""
    from egotap_amd import lib
    qkv = _rand((B * N, 3 * heads * 128), 41, -2, 2)
    got = lib.attention(qkv.cuda(), B, N, heads, precision="bf16x3")
    _close(got, _attn_ref(qkv, B, N, heads), 4e-5)
    assert torch.equal(got, lib.attention(qkv.cuda(), B, N, heads, precision="bf16x3"))


@pytest.mark.parametrize("B,N,heads", [(2, 576, 8), (1, 96, 1)])
def test_attention_bf16_plain(B, N, heads):
    from egotap_amd import lib
    qkv = _rand((B * N, 3 * heads * 128), 43, -2, 2)
    got = lib.attention(qkv.cuda(), B, N, heads, precision="bf16")
    _close(got, _attn_ref(qkv, B, N, heads, rounded=True), 2e-2)      # probabilities are rounded to bf16 too


def test_attention_bf16x3_rescale_branch_and_structured_v():
    from egotap_amd import lib
    B, N, heads, D = 1, 128, 1, 128
    qkv = _rand((N, 3 * D), 51, -0.5, 0.5)
    qkv[7, :128] = 3.0
    qkv[100, 128:256] = 3.0
    # V[key][d] = key + d/1000: every (key, d) pair is distinguishable in the output
    qkv[:, 256:] = torch.arange(N, dtype=torch.float32)[:, None] + torch.arange(128, dtype=torch.float32)[None, :] / 1000.0
    _close(lib.attention(qkv.cuda(), B, N, heads, precision="bf16x3"), _attn_ref(qkv, B, N, heads), 2e-3, rtol=2e-5)


# ---- ground-truth heatmap synthesis on the device (heatmap_synth.h)
@pytest.mark.parametrize("preset,n", [("UnrealEgo", 16), ("EgoCap", 18)])
@pytest.mark.parametrize("res", [64, 128])
def test_synth_heatmaps_matches_oracle_and_reference_fixture(preset, n, res):
    import os
    from egotap_amd import lib
    from egotap_amd.synthetic import synth_input
    from oracle import heatmap_synth_ref as R
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "heatmap_synth.npz"))
    p2l = synth_input(f"synth_p2l_{preset}", (3, n, 2), -60.0, 1080.0)
    p2r = synth_input(f"synth_p2r_{preset}", (3, n, 2), -60.0, 1080.0)
    p2l[0, 1] = [512.0, 256.0]
    p2l[0, 2] = [-100.0, 500.0]
    p3 = synth_input(f"synth_p3_{preset}", (3, n, 3), -40.0, 40.0)
    out = lib.synth_heatmaps(torch.from_numpy(p2l).cuda(), torch.from_numpy(p2r).cuda(), torch.from_numpy(p3).cuda(), preset, res)
    J = n - 1
    cat = out["cat"].cpu().numpy()
    assert cat.shape == (3, 6 * J, res, res)
    for b in range(3):
        ref, plen, theta = R.process_frame(p2l[b].astype(np.float64), p2r[b].astype(np.float64), p3[b].astype(np.float64), preset, res)
        np.testing.assert_allclose(cat[b], ref, atol=2e-6, rtol=0)
        np.testing.assert_allclose(out["gt_limb_theta"][b].cpu().numpy(), theta, atol=1e-6)
        np.testing.assert_allclose(out["gt_plength_left"][b, :J].cpu().numpy(), plen[0], rtol=1e-5)
        np.testing.assert_allclose(out["gt_plength_right"][b, J:].cpu().numpy(), plen[1], rtol=1e-5)
        np.testing.assert_allclose(cat[b, :2 * J], g[f"{preset}_{res}_{b}_pos"], atol=2e-6, rtol=0)      # the reference's own output
    assert float(cat[0, 1].max()) == 0.0 or p2l[0, 2, 0] >= 0        # the off-frame joint leaves an empty position map
