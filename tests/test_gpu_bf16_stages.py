"""Orchestration of the bf16-storage training step, stage by stage (round-2 verdict, missing 3 / weak 2).

End to end the bf16 step cannot be gated tightly against ANY oracle: two float64 emulations of this very arithmetic that differ only
in the precision their sums are accumulated in (oracle.lift_ref with round=Bf16Storage, float32 vs float64 base) are 1.0e-3 apart after
ONE ViT layer, 2.8e-3 at the encoder output and 13-15 % in the gradients behind the train-mode BatchNorm1d over 60 nearly identical
rows (tools/bf16_emu_probe.py): every bf16 rounding that flips changes its element by 2^-8 and the network's
fan-out spreads it.  What CAN be gated at one bf16 rounding is every STAGE of the real, orchestrated step against that stage's own
inputs as the GPU produced them: a tensor wired to the wrong consumer, a missing residual, a transposed weight copy, a wrong bias /
gradient slot all show up as O(1) errors here, which the 20 % end-to-end gate of test_gpu_configs.py cannot see.
The step checked is the operator composition (net.one_call_training = False), which tests/test_gpu_train_step.py proves
bit-identical to the one-call ABI the wrapper trains through."""
import math

import numpy as np
import pytest
import torch

from egotap_amd.synthetic import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
ULP = 2.0 ** -8          # one bf16 rounding: relative error <= 2^-9; a gate of 2^-8 leaves room for the fp32 accumulation order


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def _ln(x, g, b, eps=1e-12):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def _ln_bwd(x, dy, g, eps=1e-12):
    """gradient of _ln w.r.t. x for upstream dy (float64)"""
    x = x.clone().requires_grad_(True)
    (dx,) = torch.autograd.grad(_ln(x, g, torch.zeros_like(g), eps), x, dy)
    return dx


def _gelu(z):
    return 0.5 * z * (1.0 + torch.erf(z / math.sqrt(2.0)))


def _dgelu(z):
    return 0.5 * (1.0 + torch.erf(z / math.sqrt(2.0))) + z * torch.exp(-0.5 * z * z) / math.sqrt(2.0 * math.pi)


def _attn(qkv, B, N, heads, D):
    dh = D // heads
    q, k, v = (qkv[:, i * D:(i + 1) * D].view(B, N, heads, dh).transpose(1, 2) for i in range(3))
    p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), dim=-1)
    return (p @ v).transpose(1, 2).reshape(B * N, D), p


@pytest.mark.parametrize("preset,B", [("UnrealEgo", 2), ("EgoCap", 3)])
def test_every_stage_of_the_bf16_step_against_its_own_inputs(preset, B):
    from egotap_amd import networks, spec
    from egotap_amd.options import preset_defaults
    from egotap_amd.training import PoseLossFn
    p = spec.lift_preset(preset)
    net = networks.EgoTAPAutoEncoder(preset_defaults(preset), input_channel_scale=2)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(spec.lift_state_spec(p)).items()})
    net = net.cuda().train()
    net.set_precision("bf16")
    net.one_call_training = False
    net._stage_trace = tr = {}
    hm = torch.from_numpy(synth_input(f"hm_stage_{preset}", (B, p.in_channels, 64, 64))).cuda()
    gt = torch.from_numpy(synth_input(f"gt_stage_{preset}", (B, p.out_joints, 3), -1.0, 1.0)).cuda()
    pose = net(hm)[0]
    PoseLossFn.apply(net, pose, gt, 0.1, -0.01).sum().backward()
    torch.cuda.synchronize()
    Sv, W = tr["saved"], tr["W"]
    P = {k: v.detach().double() for k, v in net.named_parameters()}
    G = {k: v.grad.double() for k, v in net.named_parameters() if v.grad is not None}
    D, seq, heads = p.vit_dim, p.seq, p.vit_heads
    v_ = "pos_heatmap_encoder.vit."
    d = lambda t: t.double()          # noqa: E731
    bf = lambda t: t.to(torch.float32).to(torch.bfloat16).double()      # noqa: E731
    worst = {}

    def gate(name, got, want, tol):
        r = _rel(got, want)
        worst[name] = max(worst.get(name, 0.0), r)
        assert r <= tol, f"{name}: relative L2 {r:.3e} > {tol:.1e}"

    # ------------------------------------------------------------------------------------------------ forward
    nl = p.vit_layers
    for i, L in enumerate(Sv["layers"]):
        l = f"{v_}encoder.layer.{i}."
        a = l + "attention.attention."
        Wl = W["layers"][i]
        x = d(L["x"])
        gate("y1", d(L["y1"]), _ln(x, P[l + "layernorm_before.weight"], P[l + "layernorm_before.bias"]), ULP)
        for sidx, nme in enumerate(("query", "key", "value")):      # the bf16 weight copies really are the rounded masters, in q | k | v order
            assert torch.equal(d(Wl["qkv"][sidx * D:(sidx + 1) * D]), bf(P[a + nme + ".weight"])), nme
            assert torch.equal(d(Wl["qkv_t"][:, sidx * D:(sidx + 1) * D]), bf(P[a + nme + ".weight"]).T)
        for nme, key in (("o", "attention.output.dense"), ("up", "intermediate.dense"), ("dn", "output.dense")):
            assert torch.equal(d(Wl[nme]), bf(P[l + key + ".weight"])) and torch.equal(d(Wl[nme + "_t"]), bf(P[l + key + ".weight"]).T)
        bqkv = torch.cat([P[a + n + ".bias"] for n in ("query", "key", "value")])
        gate("qkv", d(L["qkv"]), d(L["y1"]) @ d(Wl["qkv"]).T + bqkv, ULP)
        ctx_ref, prob = _attn(d(L["qkv"]), B, seq, heads, D)
        gate("ctx", d(L["ctx"]), ctx_ref, 1.5 * ULP)                               # P is rounded to bf16 inside the kernel as well
        lse_ref = torch.logsumexp((d(L["qkv"])[:, :D].view(B, seq, heads, -1).transpose(1, 2) @
                                   d(L["qkv"])[:, D:2 * D].view(B, seq, heads, -1).transpose(1, 2).transpose(-1, -2)) / math.sqrt(D // heads), dim=-1)
        assert float((d(L["lse"]).view(B, heads, seq) - lse_ref).abs().max()) < 1e-4
        gate("xm", d(L["xm"]), x + d(L["ctx"]) @ d(Wl["o"]).T + P[l + "attention.output.dense.bias"], 2e-6)          # fp32 result: no rounding
        gate("y2", d(L["y2"]), _ln(d(L["xm"]), P[l + "layernorm_after.weight"], P[l + "layernorm_after.bias"]), ULP)
        zf = d(L["y2"]) @ d(Wl["up"]).T + P[l + "intermediate.dense.bias"]
        gate("z", d(L["z"]), zf, ULP)
        gate("hid", d(L["hid"]), _gelu(zf), ULP)
        x_next = d(Sv["layers"][i + 1]["x"]) if i + 1 < nl else d(Sv["xf"])
        gate("x_out", x_next, d(L["xm"]) + d(L["hid"]) @ d(Wl["dn"]).T + P[l + "output.dense.bias"], 2e-6)
    gate("tokens", d(Sv["tokens"]), _ln(d(Sv["xf"]), P[v_ + "layernorm.weight"], P[v_ + "layernorm.bias"]), ULP)
    assert torch.equal(d(Sv["hm_b"]), bf(Sv["hm"]))
    # fc1 of both encoders: bf16 operands, fp32 result, rows eye-major [B * T, 2048]
    T_, q = p.tokens, p.ppd
    tok = d(Sv["tokens"]).view(B, p.grid, q, p.grid, q, D).permute(0, 1, 3, 2, 4, 5).reshape(B, p.grid * p.grid, q * q * D)[:, :T_].reshape(B * T_, -1)
    gate("fc1_pos", d(Sv["pos_acts"][0]["z"]), tok @ d(W["fc1p"]).T + P["pos_heatmap_encoder.fc1.fc.bias"], 2e-6)
    J, hmsz = p.n_joints_hm, p.hm_size
    rot = d(Sv["hm_b"])[:, 2 * J:].reshape(B, 2, 2, J, hmsz * hmsz).permute(0, 1, 3, 2, 4).reshape(B * 2 * J, 2 * hmsz * hmsz)
    gate("fc1_rot", d(Sv["rot_acts"][0]["z"]), rot @ d(W["fc1r"]).T + P["rot_heatmap_encoder.fc1.fc.bias"], 2e-6)

    # ------------------------------------------------------------------------------------------------ backward
    dtok = d(tr["dtok"])
    dx_ref = _ln_bwd(d(Sv["xf"]), dtok, P[v_ + "layernorm.weight"])
    gate("dx_final", d(tr["dx_final"]), dx_ref, 2e-5)
    gate("dxb_final", d(tr["dxb_final"]), d(tr["dx_final"]), ULP)
    last = f"{v_}encoder.layer.{nl - 1}."
    gate("d output.dense.bias (final-LN column sums)", G[last + "output.dense.bias"], d(tr["dx_final"]).sum(0), 1e-4)
    gate("d layernorm.weight", G[v_ + "layernorm.weight"],
         (dtok * (d(Sv["xf"]) - d(Sv["xf"]).mean(-1, keepdim=True)) / torch.sqrt(d(Sv["xf"]).var(-1, unbiased=False, keepdim=True) + 1e-12)).sum(0), 1e-4)
    gate("d layernorm.bias", G[v_ + "layernorm.bias"], dtok.sum(0), 1e-4)
    for i in reversed(range(nl)):
        L, Wl, t = Sv["layers"][i], W["layers"][i], tr[f"L{i}"]
        l = f"{v_}encoder.layer.{i}."
        a = l + "attention.attention."
        dxb = d(t["dxb_in"])
        gate("dW output.dense", G[l + "output.dense.weight"], dxb.T @ d(L["hid"]), 1e-4)
        gate("dz", d(t["dz"]), (dxb @ d(Wl["dn_t"]).T) * _dgelu(d(L["z"])), ULP)
        dz = d(t["dz"])
        gate("dW intermediate.dense", G[l + "intermediate.dense.weight"], dz.T @ d(L["y2"]), 1e-4)
        gate("db intermediate.dense", G[l + "intermediate.dense.bias"], dz.sum(0), 1e-4)
        gate("dy2", d(t["dy2"]), dz @ d(Wl["up_t"]).T, ULP)
        dxm_ref = d(t["dx_in"]) + _ln_bwd(d(L["xm"]), d(t["dy2"]), P[l + "layernorm_after.weight"])
        gate("dxm", d(t["dxm"]), dxm_ref, 2e-5)
        gate("dxmb", d(t["dxmb"]), d(t["dxm"]), ULP)
        xm_hat = (d(L["xm"]) - d(L["xm"]).mean(-1, keepdim=True)) / torch.sqrt(d(L["xm"]).var(-1, unbiased=False, keepdim=True) + 1e-12)
        gate("d layernorm_after.weight", G[l + "layernorm_after.weight"], (d(t["dy2"]) * xm_hat).sum(0), 1e-4)
        gate("d layernorm_after.bias", G[l + "layernorm_after.bias"], d(t["dy2"]).sum(0), 1e-4)
        gate("db attention.output.dense (LN column sums)", G[l + "attention.output.dense.bias"], d(t["dxm"]).sum(0), 1e-4)
        dxmb = d(t["dxmb"])
        gate("dW attention.output.dense", G[l + "attention.output.dense.weight"], dxmb.T @ d(L["ctx"]), 1e-4)
        gate("dctx", d(t["dctx"]), dxmb @ d(Wl["o_t"]).T, ULP)
        # flash-style backward as the kernels compute it (csrc/attention_bf16s.h): P recomputed from q, k and the forward's log-sum-exp,
        # delta = rowsum(dO * O) from the STORED (bf16) O, P and dS rounded to bf16 where they become MFMA operands
        dh = D // heads
        qh, kh, vh = (d(L["qkv"])[:, j * D:(j + 1) * D].view(B, seq, heads, dh).transpose(1, 2) for j in range(3))
        doh = d(t["dctx"]).view(B, seq, heads, dh).transpose(1, 2)
        oh = d(L["ctx"]).view(B, seq, heads, dh).transpose(1, 2)
        prob = torch.exp(qh @ kh.transpose(-1, -2) / math.sqrt(dh) - d(L["lse"]).view(B, heads, seq, 1))
        delta = (doh * oh).sum(-1, keepdim=True)
        ds = prob * (doh @ vh.transpose(-1, -2) - delta) / math.sqrt(dh)
        dq_ref, dk_ref, dv_ref = bf(ds) @ kh, bf(ds).transpose(-1, -2) @ qh, bf(prob).transpose(-1, -2) @ doh
        dqkv_ref = torch.cat([g_.transpose(1, 2).reshape(B * seq, D) for g_ in (dq_ref, dk_ref, dv_ref)], dim=1)
        for sidx, nme in enumerate(("query", "key", "value")):
            sl = slice(sidx * D, (sidx + 1) * D)
            gate("dqkv " + nme, d(t["dqkv"])[:, sl], dqkv_ref[:, sl], 1.5 * ULP)
            gate("dW " + nme, G[a + nme + ".weight"], d(t["dqkv"])[:, sl].T @ d(L["y1"]), 1e-4)
            gate("db " + nme, G[a + nme + ".bias"], d(t["dqkv"])[:, sl].sum(0), 1e-4)
        gate("dy1", d(t["dy1"]), d(t["dqkv"]) @ d(Wl["qkv_t"]).T, ULP)
        gate("dx_out", d(t["dx_out"]), d(t["dxm"]) + _ln_bwd(d(L["x"]), d(t["dy1"]), P[l + "layernorm_before.weight"]), 2e-5)
        if i > 0:
            gate("dxb_out", d(t["dxb_out"]), d(t["dx_out"]), ULP)
            gate("db output.dense below (LN column sums)", G[f"{v_}encoder.layer.{i - 1}.output.dense.bias"], d(t["dx_out"]).sum(0), 1e-4)
            assert torch.equal(tr[f"L{i - 1}"]["dx_in"], t["dx_out"]) and torch.equal(tr[f"L{i - 1}"]["dxb_in"], t["dxb_out"])      # wiring
    # patch embedding: position embeddings = batch sum of dx; (bias | mask token) split by the dummy cells
    dx0 = d(tr["L0"]["dx_out"]).view(B, seq, D)
    gate("d position_embeddings", G[v_ + "embeddings.position_embeddings"].reshape(seq, D), dx0.sum(0), 1e-4)
    print({k: f"{v:.2e}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:8]})
