"""Presets and state_dict contracts of the hot path.

The key names, shapes and order are the reference's, so released checkpoints
(`*_net_AutoEncoder.pth`, `*_net_HeatMap.pth`) load unchanged:
  lifting head      model/net_architecture.py:579-677 (+ modeling_vit.py, custom_cells.py)
  heatmap estimator model/net_architecture.py:25-173 over torchvision resnet18
"""
from __future__ import annotations

import math
from dataclasses import dataclass

BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked")


@dataclass(frozen=True)
class LiftPreset:
    name: str
    n_joints_hm: int          # J: heatmaps per eye == PU chain length
    estimate_head: bool       # UnrealEgo: head joint (+ global offset) from global_mlp, output LAST
    hm_size: int = 64
    hidden: int = 128         # --ae_hidden_size
    vit_dim: int = 1024
    vit_heads: int = 8
    vit_layers: int = 3
    patch: int = 16
    pu_hidden: int = 512

    @property
    def tokens(self):          # T: heatmap tokens per sample (stereo)
        return 2 * self.n_joints_hm

    @property
    def grid(self):            # net_architecture.py:328
        return int(math.sqrt(self.tokens - 1)) + 1

    @property
    def ppd(self):             # patches per heatmap side
        return self.hm_size // self.patch

    @property
    def side(self):
        return self.grid * self.ppd

    @property
    def seq(self):
        return self.side * self.side

    @property
    def out_joints(self):
        return self.n_joints_hm + (1 if self.estimate_head else 0)

    @property
    def in_channels(self):
        return 6 * self.n_joints_hm


def lift_preset(joint_preset: str = "UnrealEgo", hm_size: int = 64, hidden: int = 128) -> LiftPreset:
    if joint_preset == "UnrealEgo":
        return LiftPreset("UnrealEgo", 15, True, hm_size, hidden)
    if joint_preset == "EgoCap":
        return LiftPreset("EgoCap", 17, False, hm_size, hidden)
    raise ValueError("joint_preset is {} which is undefined".format(joint_preset))


def _linear(pre, n_out, n_in):
    return [(pre + ".weight", (n_out, n_in)), (pre + ".bias", (n_out,))]


def _fc_block(pre, n_in, n_out):
    return _linear(pre + ".fc", n_out, n_in) + [
        (pre + ".bn.weight", (n_out,)), (pre + ".bn.bias", (n_out,)),
        (pre + ".bn.running_mean", (n_out,)), (pre + ".bn.running_var", (n_out,)),
        (pre + ".bn.num_batches_tracked", ()),
    ]


def lift_state_spec(p: LiftPreset):
    """[(key, shape)] of EgoTAPAutoEncoder.state_dict(), in the reference's order."""
    D, H = p.vit_dim, p.pu_hidden
    v = "pos_heatmap_encoder.vit."
    s = [
        (v + "embeddings.cls_token", (1, 1, D)),
        (v + "embeddings.mask_token", (1, 1, D)),
        (v + "embeddings.position_embeddings", (1, p.seq, D)),
        (v + "embeddings.patch_embeddings.projection.weight", (D, 1, p.patch, p.patch)),
        (v + "embeddings.patch_embeddings.projection.bias", (D,)),
    ]
    for i in range(p.vit_layers):
        l = f"{v}encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s += _linear(l + "attention.attention." + n, D, D)
        s += _linear(l + "attention.output.dense", D, D)
        s += _linear(l + "intermediate.dense", 4 * D, D)
        s += _linear(l + "output.dense", D, 4 * D)
        s += [(l + "layernorm_before.weight", (D,)), (l + "layernorm_before.bias", (D,)),
              (l + "layernorm_after.weight", (D,)), (l + "layernorm_after.bias", (D,))]
    s += [(v + "layernorm.weight", (D,)), (v + "layernorm.bias", (D,))]
    s += _linear(v + "pooler.dense", D, D)
    for enc, k1 in (("pos_heatmap_encoder", p.ppd * p.ppd * D), ("rot_heatmap_encoder", 2 * p.hm_size * p.hm_size)):
        s += _fc_block(enc + ".fc1", k1, 2048)
        s += _fc_block(enc + ".fc2", 2048, 512)
        s += _fc_block(enc + ".fc3", 512, p.hidden)
    x = 2 * p.hidden                       # per-joint stereo feature (left|right)
    c = "skel_sequential_layer.lstm_custom.layers."
    s += _linear(c + "0.x2f", H + x, x) + _linear(c + "0.x2h", 4 * H, x)
    s += _linear(c + "0.b2h", 4 * H, x) + _linear(c + "0.h2h", 4 * H, H)
    s += _linear(c + "1.x2f", H, H) + _linear(c + "1.x2h", 4 * H, H) + _linear(c + "1.h2h", 4 * H, H)
    s += _linear("pose_mlp.pose_fcs.0", 3, x + H)
    if p.estimate_head:
        s += _linear("global_mlp.pose_fcs.0", 6, p.n_joints_hm * H)
    return s


def is_buffer(key: str) -> bool:
    return key.rsplit(".", 1)[-1] in BUFFER_LEAVES


# keys that exist in the checkpoint but never receive a gradient / are never read on the path
# (modeling_vit.py:610 pooler output discarded; use_cls_token=False, net_architecture.py:358)
LIFT_DEAD_KEYS = (
    "pos_heatmap_encoder.vit.embeddings.cls_token",
    "pos_heatmap_encoder.vit.pooler.dense.weight",
    "pos_heatmap_encoder.vit.pooler.dense.bias",
)
