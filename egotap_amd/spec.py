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


# ----------------------------------------------------------------------------------------- heatmap estimator
HM_STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))    # torchvision resnet18: (channels, stride of first block)


def _bn2d(pre, c):
    return [(pre + ".weight", (c,)), (pre + ".bias", (c,)), (pre + ".running_mean", (c,)),
            (pre + ".running_var", (c,)), (pre + ".num_batches_tracked", ())]


HM_BLOCKS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}      # BasicBlock ResNets of torchvision (net_architecture.py:57-60)
HM_BOTTLENECK = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3)}  # Bottleneck ResNets (net_architecture.py:61-64), expansion 4


def hm_blocks(model_name: str = "resnet18"):
    """BasicBlocks per stage of the nets the one-call C forward, the bf16 modes and stage-1 training cover; raises for resnet50 / resnet101
    (Bottleneck blocks: hm_bottleneck_blocks -- fp32 eval forward composed from the operator entry points, networks.py)"""
    if model_name not in HM_BLOCKS:
        raise NotImplementedError(f"backbone {model_name!r}: only the BasicBlock ResNets run on this path ({', '.join(HM_BLOCKS)}; the shipped scripts use resnet18)")
    return HM_BLOCKS[model_name]


def hm_is_bottleneck(model_name: str) -> bool:
    return model_name in HM_BOTTLENECK


def hm_all_blocks(model_name: str = "resnet18"):
    if model_name in HM_BOTTLENECK:
        return HM_BOTTLENECK[model_name]
    return hm_blocks(model_name)


def hm_feature_scale(model_name: str = "resnet18") -> int:
    """net_architecture.py:104-111: channels of the pyramid levels relative to resnet18 (Bottleneck expansion 4)"""
    return 4 if model_name in HM_BOTTLENECK else 1


def resnet18_spec(model_name: str = "resnet18"):
    """[(key, shape)] of torchvision.models.resnet18().state_dict() (public architecture; 122 entries) -- or resnet34's (218 entries),
    resnet50's (320) and resnet101's (626): Bottleneck = conv1 1x1 -> conv2 3x3 (carries the stride, torchvision's v1.5) -> conv3 1x1 to
    4 x width, downsample in the first block of every stage (layer1's too: 64 -> 256 channels)."""
    s = [("conv1.weight", (64, 3, 7, 7))] + _bn2d("bn1", 64)
    cin = 64
    blocks = hm_all_blocks(model_name)
    bott = hm_is_bottleneck(model_name)
    for i, (c, stride) in enumerate(HM_STAGES, start=1):
        cout = 4 * c if bott else c
        for b in range(blocks[i - 1]):
            pre = f"layer{i}.{b}"
            bc_in = cin if b == 0 else cout
            if bott:
                s += [(pre + ".conv1.weight", (c, bc_in, 1, 1))] + _bn2d(pre + ".bn1", c)
                s += [(pre + ".conv2.weight", (c, c, 3, 3))] + _bn2d(pre + ".bn2", c)
                s += [(pre + ".conv3.weight", (cout, c, 1, 1))] + _bn2d(pre + ".bn3", cout)
            else:
                s += [(pre + ".conv1.weight", (c, bc_in, 3, 3))] + _bn2d(pre + ".bn1", c)
                s += [(pre + ".conv2.weight", (c, c, 3, 3))] + _bn2d(pre + ".bn2", c)
            if b == 0 and (stride != 1 or cin != cout):
                s += [(pre + ".downsample.0.weight", (cout, bc_in, 1, 1))] + _bn2d(pre + ".downsample.1", cout)
        cin = cout
    s += [("fc.weight", (1000, cin)), ("fc.bias", (1000,))]
    return s


def hm_state_spec(n_hm_per_eye: int, model_name: str = "resnet18"):
    """[(key, shape, alias_of)] of HeatMap_UnrealEgo_Shared(resnet18, stereo).state_dict() in the reference's order.

    Encoder_Block registers the ResNet and, again, its slices layer0..layer4 (net_architecture.py:58, 68-73), so
    every backbone tensor appears under two keys; alias_of names the canonical key of the shared tensor.
    n_hm_per_eye = num_heatmap + 2 * num_rot_heatmap of that net (15 for the position net, 30 for the sin/cos net).
    """
    rs = resnet18_spec(model_name)
    root = "backbone.backbone."
    out = [(root + "backbone." + k, shp, None) for k, shp in rs]

    def dup(prefix_new, prefix_old):
        for k, shp in rs:
            if k.startswith(prefix_old):
                out.append((root + prefix_new + k[len(prefix_old):], shp, root + "backbone." + k))

    dup("layer0.0.", "conv1.")
    dup("layer0.1.", "bn1.")
    dup("layer1.1.", "layer1.")
    dup("layer2.", "layer2.")
    dup("layer3.", "layer3.")
    dup("layer4.", "layer4.")
    a = "after_backbone."

    def conv(name, cout, cin, k):
        return [(a + name + ".weight", (cout, cin, k, k), None), (a + name + ".bias", (cout,), None)]

    f = 2 * hm_feature_scale(model_name)            # feature_scale * input_channel_scale (net_architecture.py:113)
    out += conv("layer1_1x1.0", 64 * f, 64 * f, 1) + conv("layer2_1x1.0", 128 * f, 128 * f, 1)
    out += conv("layer3_1x1.0", 258 * f, 256 * f, 1) + conv("layer4_1x1.0", 512 * f, 512 * f, 1)
    out += conv("conv_up3.0", 512 * f, 258 * f + 512 * f, 3) + conv("conv_up2.0", 256 * f, 128 * f + 512 * f, 3)
    out += conv("conv_up1.0", 256 * f, 64 * f + 256 * f, 3)
    out += conv("conv_heatmap", 2 * n_hm_per_eye, 256 * f, 1)
    return out
