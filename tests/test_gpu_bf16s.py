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


def test_gemm_nt_f32_out():
    from egotap_amd import bf16s
    M, N, K = 930, 2048, 1024
    x, w, b = _rand((M, K), 31).bfloat16(), (_rand((N, K), 32) / math.sqrt(K)).bfloat16(), _rand((N,), 33)
    out = bf16s.gemm_nt(x.cuda(), w.cuda(), b.cuda(), epi="f32")
    ref = x.double() @ w.double().T + b.double()
    assert out.dtype == torch.float32 and float((out.double().cpu() - ref).abs().max()) < 2e-6 * math.sqrt(K) + 1e-6
