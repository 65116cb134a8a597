"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of the reference's heatmap -> 3D lifting head, written as flat
functions over a ``state_dict`` (the reference's own key names, SURVEY.md
Appendix B) with plain torch CPU tensor ops.  Pinned against golden vectors
captured from the reference's own modules (tests/golden/lift_fwd_*.npz,
pu_chain_*.npz, fcblock.npz, loss_*.npz; generator tools/make_golden.py), see
tests/test_oracle_golden.py.

Only ``tests/``, ``__graft_entry__.smoke()``, ``bench.py``'s cpu_baseline leg and the
measurement probes under ``tools/`` may import this file.

``round=Bf16Storage`` (optional argument of the forward / training functions) rounds
to bf16 at the tensors the HIP bf16-storage mode keeps in bf16: an emulation of THIS
repo's reduced-precision arithmetic for tighter test gates, not reference behaviour;
``round=None`` is the pinned restatement.

Reference lines restated (all under /root/reference):
  model/net_architecture.py:682-758   EgoTAPAutoEncoder.forward  -> lift_forward
  model/net_architecture.py:370-415   PatchedHeatmapFeatureExtractorViT.forward -> pos_encoder
  model/net_architecture.py:263-274   HeatmapFeatureExtractorFC.forward -> rot_encoder
  model/net_architecture.py:513-576 + model/custom_cells.py:94-197  SkelNet/PU -> pu_chain
  model/modeling_vit.py:128-157, 195, 226-252, 271, 319-344, 366-384, 608-609 -> vit_*
  model/network_utils.py:123-142      make_fc_layer -> fc_block
  utils/loss.py:54-85                 LossFuncCosSim / LossFuncMPJPE -> loss_*
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

UE_PARENTS = [0, 0, 1, 1, 2, 3, 4, 5, 2, 3, 8, 9, 10, 11, 12, 13]            # utils/util.py:51
EC_PARENTS = [0, 0, 1, 2, 3, 4, 1, 6, 7, 8, 2, 10, 11, 12, 6, 14, 15, 16]   # utils/util.py:52


@dataclass(frozen=True)
class LiftPreset:
    name: str
    n_joints_hm: int          # J: heatmaps per eye (= PU chain length)
    estimate_head: bool       # UE: extra head joint from global_mlp, appended LAST
    hm_size: int = 64
    hidden: int = 128         # ae_hidden_size
    vit_dim: int = 1024
    vit_heads: int = 8
    vit_layers: int = 3
    patch: int = 16

    @property
    def tokens(self):          # T = 2J (stereo)
        return 2 * self.n_joints_hm

    @property
    def grid(self):            # heatmaps per side of the tiled ViT image (net_architecture.py:328)
        return int(math.sqrt(self.tokens - 1)) + 1

    @property
    def ppd(self):             # patches per heatmap side
        return self.hm_size // self.patch

    @property
    def side(self):            # patches per side of the ViT image
        return self.grid * self.ppd

    @property
    def seq(self):
        return self.side * self.side

    @property
    def out_joints(self):
        return self.n_joints_hm + (1 if self.estimate_head else 0)

    @property
    def in_channels(self):     # 2J position + 4J sin/cos
        return 6 * self.n_joints_hm


UE = LiftPreset("UnrealEgo", 15, True)
EC = LiftPreset("EgoCap", 17, False)


def preset_by_name(name: str, hm_size: int = 64) -> LiftPreset:
    base = {"UnrealEgo": UE, "EgoCap": EC}[name]
    return LiftPreset(base.name, base.n_joints_hm, base.estimate_head, hm_size)


# ---------------------------------------------------------------------------
def layer_norm(x, w, b, eps=1e-12):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


# ---------------------------------------------------------------------------
# Optional emulation of the bf16-storage mode (DESIGN.md section 2 "bf16-storage mode"; what --use_amp maps to).  The reference has
# no such arithmetic (it autocasts to fp16): this hook exists so that the HIP bf16 path can be gated against an oracle that rounds
# at the SAME tensors (~1e-3) instead of against exact float64 with a 20 % allowance.  round=None (the default everywhere) leaves
# every function below exactly as pinned by the golden vectors.
def bf16_round(x):
    return x.to(torch.float32).to(torch.bfloat16).to(x.dtype)


class _Q(torch.autograd.Function):
    """identity whose forward and / or backward value is rounded to bf16"""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return bf16_round(x) if fwd else x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return (bf16_round(g) if ctx.bwd else g), None, None


class _GeluStored(torch.autograd.Function):
    """SEpiGeluSave / SEpiGeluGrad (csrc/gemm_bf16s.h): hid = bf16(GELU(z)) from the fp32 pre-activation; the backward evaluates
    GELU' at the STORED pre-activation bf16(z) and writes dz as bf16"""

    @staticmethod
    def forward(ctx, z):
        ctx.save_for_backward(bf16_round(z))
        return bf16_round(gelu_erf(z))

    @staticmethod
    def backward(ctx, g):
        (zs,) = ctx.saved_tensors
        d = 0.5 * (1.0 + torch.erf(zs / math.sqrt(2.0))) + zs * torch.exp(-0.5 * zs * zs) / math.sqrt(2.0 * math.pi)
        return bf16_round(g * d)


class Bf16Storage:
    """round= hook: where the bf16-storage step keeps bf16 in HBM.
      act : tensor stored as bf16 whose gradient is stored as bf16 too (LayerNorm outputs, qkv, ctx, tokens)
      fwd : forward value rounded, gradient untouched (probabilities entering P.V; GEMM inputs that get no gradient)
      bwd : gradient rounded, value untouched (the bf16 copy of dx a GEMM branch consumes; dS inside attention)
      w   : per-step bf16 copy of a weight, gradient passed straight through to the fp32 master
      gelu: the MLP's saved pre-activation / activation pair"""
    act = staticmethod(lambda x: _Q.apply(x, True, True))
    fwd = staticmethod(lambda x: _Q.apply(x, True, False))
    bwd = staticmethod(lambda x: _Q.apply(x, False, True))
    w = staticmethod(lambda x: _Q.apply(x, True, False))
    gelu = staticmethod(_GeluStored.apply)


def fc_block(x, sd, prefix, training=False, momentum=0.1, eps=1e-5, round=None, round_input=False, lrelu=None):
    """LeakyReLU_0.2(BatchNorm1d(x W^T + b)); network_utils.py:123-142.

    training=True uses batch statistics and returns the updated running stats
    (unbiased variance, momentum 0.1) as torch's BatchNorm1d does.

    lrelu (tests only; None everywhere else = the pinned function): {"masks": {prefix: bool [R, N]}, "pre": {}}.  The gradient of
    this block is discontinuous where the LeakyReLU input crosses zero, and two fp32 implementations of the matrix product in front of
    it disagree about the sign of an input within rounding of zero (a few elements per step at fc1's 16384-long dot products): with
    a mask given, the positive branch is taken exactly where the mask says (the function "as the implementation under test evaluated
    it"); "pre" receives the float64 LeakyReLU input so that the test can check that the two only differ where it is ~0.
    """
    if round is not None:       # fc1 of both encoders: bf16 operands; dz is rounded for the weight- / input-gradient GEMMs, the bias sees fp32
        xin = round.fwd(x) if round_input else x
        z = round.bwd(xin @ round.w(sd[prefix + ".fc.weight"]).T) + sd[prefix + ".fc.bias"]
    else:
        z = x @ sd[prefix + ".fc.weight"].T + sd[prefix + ".fc.bias"]
    g, beta = sd[prefix + ".bn.weight"], sd[prefix + ".bn.bias"]
    if training:
        mean = z.mean(dim=0)
        var_b = ((z - mean) ** 2).mean(dim=0)
        n = z.shape[0]
        new_rm = (1 - momentum) * sd[prefix + ".bn.running_mean"] + momentum * mean
        new_rv = (1 - momentum) * sd[prefix + ".bn.running_var"] + momentum * var_b * n / (n - 1)
        y = (z - mean) / torch.sqrt(var_b + eps) * g + beta
        if lrelu is not None:
            lrelu["pre"][prefix] = y.detach()
            if prefix in lrelu["masks"]:
                return torch.where(lrelu["masks"][prefix], y, 0.2 * y), (new_rm, new_rv)
        return F.leaky_relu(y, 0.2), (new_rm, new_rv)
    mean, var = sd[prefix + ".bn.running_mean"], sd[prefix + ".bn.running_var"]
    y = (z - mean) / torch.sqrt(var + eps) * g + beta
    return F.leaky_relu(y, 0.2)


def tile_to_patches(pos_hm, p: LiftPreset):
    """[B,T,hm,hm] -> ([B, seq, patch*patch], dummy_mask[seq]).

    Heatmap i sits at grid cell (i // grid, i % grid) of a (grid*hm)^2 image whose
    unused cells are zero (net_architecture.py:375-383); the image is cut in
    row-major 16x16 patches (modeling_vit.py:195).
    """
    B, T = pos_hm.shape[:2]
    G, hm, ps, S = p.grid, p.hm_size, p.patch, p.side
    cells = torch.zeros(B, G * G, hm, hm, dtype=pos_hm.dtype)
    cells[:, :T] = pos_hm
    img = cells.view(B, G, G, hm, hm).permute(0, 1, 3, 2, 4).reshape(B, G * hm, G * hm)
    patches = img.view(B, S, ps, S, ps).permute(0, 1, 3, 2, 4).reshape(B, S * S, ps * ps)
    cell_of_patch = (torch.arange(S)[:, None] // p.ppd) * G + (torch.arange(S)[None, :] // p.ppd)
    dummy = (cell_of_patch >= T).reshape(-1)
    return patches, dummy


def vit_embed(pos_hm, sd, p: LiftPreset, pre="pos_heatmap_encoder.vit.", round=None):
    patches, dummy = tile_to_patches(pos_hm, p)
    w = sd[pre + "embeddings.patch_embeddings.projection.weight"].reshape(p.vit_dim, -1)
    b = sd[pre + "embeddings.patch_embeddings.projection.bias"]
    if round is not None:       # operands rounded in registers, fp32 result (round-1 kernels in this mode)
        emb = round.bwd(round.fwd(patches) @ round.w(w).T) + b
    else:
        emb = patches @ w.T + b
    mask_tok = sd[pre + "embeddings.mask_token"].reshape(-1)
    emb = torch.where(dummy[None, :, None], mask_tok[None, None, :], emb)
    return emb + sd[pre + "embeddings.position_embeddings"]


def vit_layer(x, sd, pre, heads, round=None):
    if round is not None:
        return _vit_layer_bf16(x, sd, pre, heads, round)
    B, N, D = x.shape
    dh = D // heads
    h = layer_norm(x, sd[pre + "layernorm_before.weight"], sd[pre + "layernorm_before.bias"])
    a = pre + "attention.attention."
    q = (h @ sd[a + "query.weight"].T + sd[a + "query.bias"]).view(B, N, heads, dh).transpose(1, 2)
    k = (h @ sd[a + "key.weight"].T + sd[a + "key.bias"]).view(B, N, heads, dh).transpose(1, 2)
    v = (h @ sd[a + "value.weight"].T + sd[a + "value.bias"]).view(B, N, heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    ctx = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, N, D)
    x = x + ctx @ sd[pre + "attention.output.dense.weight"].T + sd[pre + "attention.output.dense.bias"]
    h = layer_norm(x, sd[pre + "layernorm_after.weight"], sd[pre + "layernorm_after.bias"])
    h = gelu_erf(h @ sd[pre + "intermediate.dense.weight"].T + sd[pre + "intermediate.dense.bias"])
    return x + h @ sd[pre + "output.dense.weight"].T + sd[pre + "output.dense.bias"]


def _vit_layer_bf16(x, sd, pre, heads, r):
    """the same layer with the bf16-storage step's rounding points (training.py LiftTrainBf16Fn / egotap_lift_forward_train):
    fp32 residual stream; y1, qkv, ctx, y2, z, hid in bf16; every large product on bf16 operands; P and dS rounded inside attention"""
    B, N, D = x.shape
    dh = D // heads
    h = r.act(layer_norm(x, sd[pre + "layernorm_before.weight"], sd[pre + "layernorm_before.bias"]))
    a = pre + "attention.attention."
    lin = lambda t, n: r.act(t @ r.w(sd[a + n + ".weight"]).T + sd[a + n + ".bias"])      # noqa: E731   (bias gradient = column sums of bf16 dqkv)
    q, k, v = (lin(h, n).view(B, N, heads, dh).transpose(1, 2) for n in ("query", "key", "value"))
    s = r.bwd((q @ k.transpose(-1, -2)) / math.sqrt(dh))
    ctx = r.act((r.fwd(torch.softmax(s, dim=-1)) @ v).transpose(1, 2).reshape(B, N, D))
    x = x + r.bwd(ctx @ r.w(sd[pre + "attention.output.dense.weight"]).T) + sd[pre + "attention.output.dense.bias"]
    h = r.act(layer_norm(x, sd[pre + "layernorm_after.weight"], sd[pre + "layernorm_after.bias"]))
    h = r.gelu(h @ r.w(sd[pre + "intermediate.dense.weight"]).T + sd[pre + "intermediate.dense.bias"])
    return x + r.bwd(h @ r.w(sd[pre + "output.dense.weight"]).T) + sd[pre + "output.dense.bias"]


def pos_encoder(pos_hm, sd, p: LiftPreset, trace=None, training=False, round=None, lrelu=None):
    """[B,T,hm,hm] -> [B*T, hidden]  (T tokens in [L_1..L_J, R_1..R_J] order)."""
    pre = "pos_heatmap_encoder."
    x = vit_embed(pos_hm, sd, p, round=round)
    if trace is not None:
        trace["emb"] = x
    for i in range(p.vit_layers):
        x = vit_layer(x, sd, f"{pre}vit.encoder.layer.{i}.", p.vit_heads, round=round)
        if trace is not None:
            trace[f"layer{i}"] = x
    x = layer_norm(x, sd[pre + "vit.layernorm.weight"], sd[pre + "vit.layernorm.bias"])
    if round is not None:
        x = round.act(x)
    if trace is not None:
        trace["final_ln"] = x
    B, T, G, q, S, D = pos_hm.shape[0], p.tokens, p.grid, p.ppd, p.side, p.vit_dim
    # heatmap i <- its q x q patches, flattened (patch-row, patch-col, channel); net_architecture.py:388-402
    grid = x.view(B, G, q, G, q, D).permute(0, 1, 3, 2, 4, 5).reshape(B, G * G, q * q * D)
    z = grid[:, :T].reshape(B * T, q * q * D)
    stats = {}
    for name in ("fc1", "fc2", "fc3"):
        z = fc_block(z, sd, pre + name, training, round=round if name == "fc1" else None, lrelu=lrelu)
        if training:
            z, stats[pre + name] = z
    return (z, stats) if training else z


def rot_relayout(rot_hm, p: LiftPreset):
    """[B,4J,hm,hm] (L_cos[J], L_sin[J], R_cos[J], R_sin[J]) -> [B*2J, 2*hm*hm]
    rows ordered [L_1..L_J, R_1..R_J], each row = (cos map | sin map); net_architecture.py:690-694."""
    B, J, hm = rot_hm.shape[0], p.n_joints_hm, p.hm_size
    return rot_hm.view(B, 2, 2, J, hm * hm).permute(0, 1, 3, 2, 4).reshape(B * 2 * J, 2 * hm * hm)


def rot_encoder(rot_hm, sd, p: LiftPreset, training=False, round=None, lrelu=None):
    z = rot_relayout(rot_hm, p)
    stats = {}
    for name in ("fc1", "fc2", "fc3"):
        z = fc_block(z, sd, "rot_heatmap_encoder." + name, training, round=round if name == "fc1" else None, round_input=True, lrelu=lrelu)
        if training:
            z, stats["rot_heatmap_encoder." + name] = z
    return (z, stats) if training else z


def stereo_interleave(z, B, p: LiftPreset):
    """[B*2J, h] (eye-major) -> [B, J, 2h] with joint j = [left_j | right_j]; net_architecture.py:699-705."""
    J, h = p.n_joints_hm, z.shape[-1]
    return z.view(B, 2, J, h).transpose(1, 2).reshape(B, J, 2 * h)


def _pu_cell(x, b, h, c, sd, pre, hidden):
    f = x @ sd[pre + "x2f.weight"].T + sd[pre + "x2f.bias"]
    h = torch.sigmoid(f[:, :hidden]) * h
    g = x @ sd[pre + "x2h.weight"].T + sd[pre + "x2h.bias"] + h @ sd[pre + "h2h.weight"].T + sd[pre + "h2h.bias"]
    if b is not None:
        b = torch.sigmoid(f[:, hidden:]) * b
        g = g + b @ sd[pre + "b2h.weight"].T + sd[pre + "b2h.bias"]
    fg, ig, cg, og = g.chunk(4, dim=1)     # forget, input, cell, output: custom_cells.py:109
    c = c * torch.sigmoid(fg) + torch.sigmoid(ig) * torch.tanh(cg)
    return torch.sigmoid(og) * torch.tanh(c), c


def pu_chain(x_seq, b_seq, sd, pre="skel_sequential_layer.lstm_custom.layers.", hidden=512, tree_parents=None):
    """x_seq, b_seq: [J, B, 256] -> [J, B, 512].

    The reference's PropagationUnit writes its new state into the caller's
    tensors (custom_cells.py:190-191), so SkelNet's per-joint state lists alias
    ONE tensor and the recurrence is a chain over joint index, a 2-layer
    modified LSTM.  ``tree_parents`` (kinematic parent list)
    computes what a real tree propagation would give -- the WRONG answer, kept
    only so a test can show the fixtures discriminate the two.
    """
    J, B = x_seq.shape[:2]
    zero = torch.zeros(B, hidden, dtype=x_seq.dtype)
    if tree_parents is None:
        h0, c0, h1, c1 = zero, zero, zero, zero
        out = []
        for t in range(J):
            h0, c0 = _pu_cell(x_seq[t], b_seq[t], h0, c0, sd, pre + "0.", hidden)
            h1, c1 = _pu_cell(h0, None, h1, c1, sd, pre + "1.", hidden)
            out.append(h1)
        return torch.stack(out)
    states = [(zero, zero, zero, zero)]
    out = []
    for i in range(1, len(tree_parents)):
        h0, c0, h1, c1 = states[tree_parents[i]]
        h0, c0 = _pu_cell(x_seq[i - 1], b_seq[i - 1], h0, c0, sd, pre + "0.", hidden)
        h1, c1 = _pu_cell(h0, None, h1, c1, sd, pre + "1.", hidden)
        states.append((h0, c0, h1, c1))
        out.append(h1)
    return torch.stack(out)


def pose_head(pos_j, skel, sd, p: LiftPreset):
    """pos_j [B,J,256], skel [J,B,512] -> [B, out_joints, 3]; net_architecture.py:732-751."""
    B, J = pos_j.shape[0], p.n_joints_hm
    skel_b = skel.transpose(0, 1)                                  # [B,J,512]
    feat = torch.cat([pos_j, skel_b], dim=-1)                      # [B,J,768]
    pose = feat @ sd["pose_mlp.pose_fcs.0.weight"].T + sd["pose_mlp.pose_fcs.0.bias"]
    if p.estimate_head:
        other = skel_b.reshape(B, -1) @ sd["global_mlp.pose_fcs.0.weight"].T + sd["global_mlp.pose_fcs.0.bias"]
        pose = pose + other[:, None, :3]
        pose = torch.cat([pose, other[:, None, 3:]], dim=1)         # head joint LAST
    return pose


def lift_forward(hm, sd, p: LiftPreset, trace=None, round=None):
    """hm [B, 6J, hm, hm] -> pose [B, out_joints, 3] (eval mode).  round: None (the reference's arithmetic) or Bf16Storage."""
    B, J = hm.shape[0], p.n_joints_hm
    pos = pos_encoder(hm[:, : 2 * J], sd, p, trace, round=round)
    rot = rot_encoder(hm[:, 2 * J:], sd, p, round=round)
    pos_j = stereo_interleave(pos, B, p)
    rot_j = stereo_interleave(rot, B, p)
    skel = pu_chain(pos_j.transpose(0, 1), rot_j.transpose(0, 1), sd)
    pose = pose_head(pos_j, skel, sd, p)
    if trace is not None:
        trace["pos_embed"] = pos.reshape(B, -1)
        trace["rot_embed"] = rot.reshape(B, -1)
        trace["skel_embed"] = skel
    return pose


# ---------------------------------------------------------------------------
def loss_mpjpe(pred, gt):
    return torch.linalg.norm(gt - pred, dim=-1).mean()


def loss_cos_sim(pred, gt, p: LiftPreset, eps=1e-8):
    """Sum over bones of cos(bone_pred, bone_gt), mean over batch; utils/loss.py:54-77."""
    parents = UE_PARENTS if p.estimate_head else EC_PARENTS
    if not p.estimate_head:      # EgoCap: zero root prepended, first bone dropped afterwards
        z = torch.zeros(pred.shape[0], 1, 3, dtype=pred.dtype)
        pred, gt = torch.cat([z, pred], 1), torch.cat([z, gt], 1)
    bp = (pred - pred[:, parents])[:, 1:]
    bg = (gt - gt[:, parents])[:, 1:]
    # nn.CosineSimilarity semantics: x.y / (max(|x|, eps) * max(|y|, eps))
    cos = (bp * bg).sum(-1) / (bp.norm(dim=-1).clamp_min(eps) * bg.norm(dim=-1).clamp_min(eps))
    if not p.estimate_head:
        cos = cos[:, 1:]
    return cos.sum(dim=1).mean()


def lift_forward_train(hm, sd, p: LiftPreset, round=None, lrelu=None):
    """Training-mode forward (BatchNorm1d batch statistics in the six FC blocks): returns (pose, {bn prefix: (new running
    mean, new running var)}).  Differentiable with torch autograd w.r.t. every tensor of sd that requires grad."""
    B, J = hm.shape[0], p.n_joints_hm
    pos, st1 = pos_encoder(hm[:, : 2 * J], sd, p, None, training=True, round=round, lrelu=lrelu)
    rot, st2 = rot_encoder(hm[:, 2 * J:], sd, p, training=True, round=round, lrelu=lrelu)
    pos_j = stereo_interleave(pos, B, p)
    rot_j = stereo_interleave(rot, B, p)
    skel = pu_chain(pos_j.transpose(0, 1), rot_j.transpose(0, 1), sd)
    st1.update(st2)
    return pose_head(pos_j, skel, sd, p), st1


def train_step(hm, gt, sd, p: LiftPreset, lam_mpjpe=0.1, lam_cos=-0.01, lr=1e-3, eps=1e-4, wd=0.0, round=None, lrelu=None):
    """One optimisation step of egotap_autoencoder_model.py:299-323 (fp32, no AMP): forward in train mode, loss =
    lam_mpjpe * MPJPE + lam_cos * lam_mpjpe * CosSim, backward, AdamW (network.py:72-78; betas 0.9 / 0.999).
    Returns dict(pose, loss_pose, loss_cos_sim, grads{key}, new_params{key}, bn{prefix: (rm, rv)})."""
    dead = ("cls_token", "pooler.dense")
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and not k.endswith(("running_mean", "running_var")) and not any(d in k for d in dead)}
    full = dict(sd)
    full.update(leaves)
    pose, bn = lift_forward_train(hm, full, p, round=round, lrelu=lrelu)
    lp = loss_mpjpe(pose, gt) * lam_mpjpe
    lc = loss_cos_sim(pose, gt, p) * lam_cos * lam_mpjpe
    grads = dict(zip(leaves.keys(), torch.autograd.grad(lp + lc, list(leaves.values()), allow_unused=True)))
    new = {}
    for k, g in grads.items():
        if g is None:
            continue
        m = 0.1 * g
        v = 0.001 * g * g
        w = leaves[k].detach() * (1 - lr * wd)
        new[k] = w - (lr / (1 - 0.9)) * m / (v.sqrt() / math.sqrt(1 - 0.999) + eps)
    return dict(pose=pose.detach(), loss_pose=lp.detach(), loss_cos_sim=lc.detach(), grads=grads, new_params=new, bn=bn)


def procrustes_align(s1, s2):
    """Batched similarity transform of s1 [B,J,3] onto s2 (utils/util.py:328-379), for PA-MPJPE."""
    x1, x2 = s1.transpose(1, 2), s2.transpose(1, 2)
    mu1, mu2 = x1.mean(-1, keepdim=True), x2.mean(-1, keepdim=True)
    y1, y2 = x1 - mu1, x2 - mu2
    var1 = (y1 ** 2).sum(dim=(1, 2))
    k = y1 @ y2.transpose(1, 2)
    u, _, vh = torch.linalg.svd(k)
    v = vh.transpose(1, 2)
    z = torch.eye(3, dtype=s1.dtype).repeat(s1.shape[0], 1, 1)
    z[:, 2, 2] = torch.sign(torch.det(u @ v.transpose(1, 2)))
    r = v @ z @ u.transpose(1, 2)
    scale = torch.diagonal(r @ k, dim1=1, dim2=2).sum(-1) / var1
    t = mu2 - scale[:, None, None] * (r @ mu1)
    return (scale[:, None, None] * (r @ x1) + t).transpose(1, 2)


def procrustes_align_batch_axes(s1, s2):
    """What utils/util.py:328-379 computes for a BATCH of 2 or 3 frames: line 337 tests S1.shape[0] against 3 and 2 (meant for
    unbatched 3 x N / 2 x N points) and skips the transpose, so s1 [B,J,3] is read as J "coordinates" x 3 "points": means over the
    three columns, a J x J outer product K of rank <= 2, its SVD, no transpose back.  Restated as written (float64 in the tests)."""
    mu1, mu2 = s1.mean(-1, keepdim=True), s2.mean(-1, keepdim=True)
    y1, y2 = s1 - mu1, s2 - mu2
    var1 = (y1 ** 2).sum(dim=(1, 2))
    k = y1 @ y2.transpose(1, 2)
    u, _, vh = torch.linalg.svd(k)
    v = vh.transpose(1, 2)
    z = torch.eye(k.shape[1], dtype=s1.dtype).repeat(s1.shape[0], 1, 1)
    z[:, -1, -1] = torch.sign(torch.det(u @ v.transpose(1, 2)))
    r = v @ z @ u.transpose(1, 2)
    scale = torch.diagonal(r @ k, dim1=1, dim2=2).sum(-1) / var1
    t = mu2 - scale[:, None, None] * (r @ mu1)
    return scale[:, None, None] * (r @ s1) + t


def to_torch_sd(np_sd, dtype=torch.float32):
    out = {}
    for k, v in np_sd.items():
        t = torch.from_numpy(v)
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out
