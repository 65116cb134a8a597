"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of the reference's stereo heatmap estimator HeatMap_UnrealEgo_Shared
(/root/reference/model/net_architecture.py:25-173): weight-shared ResNet-18 on the left and right image,
channel concat of the two feature pyramids, U-Net decoder (1x1 convs, 3x bilinear x2 upsample with
align_corners=True, 3x 3x3 conv + ReLU, final 1x1 conv).

The ResNet-18 itself is THIRD-PARTY arithmetic the reference takes from torchvision (not installed here, version
unpinned: requirements.txt just says `torchvision`).  It is restated from the public definition (conv7x7/2 - BN -
ReLU - maxpool3x3/2 - 4 stages x 2 BasicBlocks, stride-2 + 1x1 downsample at stages 2-4, BN eps 1e-5) with
torchvision's state_dict names.  Parity of the backbone against torchvision is therefore UNPINNED; everything after
the backbone is pinned by tests/golden/hm_full_*.npz (the reference's own AfterBackbone and glue code run on top of
`ResNet18Holder`, which tools/make_golden.py installs as the torchvision stub).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))     # (channels, stride of the first block)


# ----------------------------------------------------------------------------- torchvision stand-in (module form)
class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class ResNet18Holder(nn.Module):
    """Module with torchvision.models.resnet18's children order and state_dict names (conv1, bn1, relu, maxpool,
    layer1..4, avgpool, fc), which is all Encoder_Block relies on (net_architecture.py:68-73)."""

    def __init__(self, blocks=(2, 2, 2, 2)):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (c, s) in enumerate(STAGES, start=1):
            setattr(self, f"layer{i}", nn.Sequential(_BasicBlock(cin, c, s), *[_BasicBlock(c, c, 1) for _ in range(blocks[i - 1] - 1)]))
            cin = c
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)

    def forward(self, x):   # never used by the reference (it calls the children itself)
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        for i in range(1, 5):
            x = getattr(self, f"layer{i}")(x)
        return self.fc(torch.flatten(self.avgpool(x), 1))


# ----------------------------------------------------------------------------- functional restatement
_TRAIN = {"on": False, "stats": None}     # train-mode BatchNorm (batch statistics) for the stage-1 training oracle


def _bn(x, sd, pre, eps=1e-5):
    w, b, m, v = sd[pre + ".weight"], sd[pre + ".bias"], sd[pre + ".running_mean"], sd[pre + ".running_var"]
    if _TRAIN["on"]:
        # the backbone runs once per eye (net_architecture.py:45-50): the right eye's update lands on top of the left eye's
        st = _TRAIN["stats"]
        rm = st.get(pre + ".running_mean", m).detach().clone()
        rv = st.get(pre + ".running_var", v).detach().clone()
        y = F.batch_norm(x, rm, rv, w, b, True, 0.1, eps)          # nn.BatchNorm2d in train mode (momentum 0.1, unbiased running var)
        _TRAIN["stats"][pre + ".running_mean"], _TRAIN["stats"][pre + ".running_var"] = rm, rv
        return y
    return (x - m[None, :, None, None]) / torch.sqrt(v[None, :, None, None] + eps) * w[None, :, None, None] + b[None, :, None, None]


def _block(x, sd, pre, stride):
    idt = x
    if (pre + ".downsample.0.weight") in sd:
        idt = _bn(F.conv2d(x, sd[pre + ".downsample.0.weight"], None, stride), sd, pre + ".downsample.1")
    if (pre + ".conv3.weight") in sd:
        # torchvision's Bottleneck (resnet50 / resnet101, net_architecture.py:61-64), public definition: 1x1 -> 3x3 carrying the stride
        # ("ResNet v1.5") -> 1x1 to 4 x width, each followed by BatchNorm; ReLU after the first two and after the residual sum.  UNPINNED
        # like the BasicBlock (torchvision is not installed here).
        y = F.relu(_bn(F.conv2d(x, sd[pre + ".conv1.weight"]), sd, pre + ".bn1"))
        y = F.relu(_bn(F.conv2d(y, sd[pre + ".conv2.weight"], None, stride, 1), sd, pre + ".bn2"))
        y = _bn(F.conv2d(y, sd[pre + ".conv3.weight"]), sd, pre + ".bn3")
        return F.relu(y + idt)
    y = F.relu(_bn(F.conv2d(x, sd[pre + ".conv1.weight"], None, stride, 1), sd, pre + ".bn1"))
    y = _bn(F.conv2d(y, sd[pre + ".conv2.weight"], None, 1, 1), sd, pre + ".bn2")
    return F.relu(y + idt)


def resnet18_pyramid(x, sd, pre="backbone.backbone.backbone."):
    """[n,3,H,W] -> (layer0 64@H/2, layer1 64@H/4, layer2 128@H/8, layer3 256@H/16, layer4 512@H/32), eval-mode BN."""
    l0 = F.relu(_bn(F.conv2d(x, sd[pre + "conv1.weight"], None, 2, 3), sd, pre + "bn1"))
    y = F.max_pool2d(l0, 3, 2, 1)
    outs = [l0]
    for i, (c, s) in enumerate(STAGES, start=1):
        b = 0
        while f"{pre}layer{i}.{b}.conv1.weight" in sd:      # 2 BasicBlocks per stage in resnet18, (3, 4, 6, 3) in resnet34; Bottlenecks: resnet50 (3, 4, 6, 3), resnet101 (3, 4, 23, 3)
            y = _block(y, sd, f"{pre}layer{i}.{b}", s if b == 0 else 1)
            b += 1
        outs.append(y)
    return outs


def _convrelu(x, sd, pre, pad):
    return F.relu(F.conv2d(x, sd[pre + ".0.weight"], sd[pre + ".0.bias"], 1, pad))


def _up(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


def after_backbone(pl, pr, sd, pre="after_backbone.", trace=None):
    """pl, pr: left/right pyramids (lists layer0..layer4) -> [B, 2*n_hm, H/4, W/4]; net_architecture.py:139-173."""
    l1, l2, l3, l4 = (torch.cat([a, b], dim=1) for a, b in zip(pl[1:], pr[1:]))
    x = _up(_convrelu(l4, sd, pre + "layer4_1x1", 0))
    x = _convrelu(torch.cat([x, _convrelu(l3, sd, pre + "layer3_1x1", 0)], 1), sd, pre + "conv_up3", 1)
    if trace is not None:
        trace["conv_up3"] = x
    x = _up(x)
    x = _convrelu(torch.cat([x, _convrelu(l2, sd, pre + "layer2_1x1", 0)], 1), sd, pre + "conv_up2", 1)
    if trace is not None:
        trace["conv_up2"] = x
    x = _up(x)
    x = _convrelu(torch.cat([x, _convrelu(l1, sd, pre + "layer1_1x1", 0)], 1), sd, pre + "conv_up1", 1)
    if trace is not None:
        trace["conv_up1"] = x
    return F.conv2d(x, sd[pre + "conv_heatmap.weight"], sd[pre + "conv_heatmap.bias"])


def hm_forward(left, right, sd, trace=None):
    """HeatMap_UnrealEgo_Shared.forward(left, right): [B,3,H,W] x2 -> [B, 2*n_hm, H/4, W/4]."""
    pl = resnet18_pyramid(left, sd)
    pr = resnet18_pyramid(right, sd)
    if trace is not None:
        for i, t in enumerate(pr):
            trace[f"pyr{i}"] = t
    return after_backbone(pl, pr, sd, trace=trace)


def hm_train_step(left, right, gt, plen, sd, lam=1.0):
    """One stage-1 training forward + backward (model/heatmap_shared_model.py:98-151): train-mode forward, loss =
    lam * (MSE(left half) + MSE(right half)) with the limb maps divided by sqrt(gt_plength) first (plen [B, C] or None).
    Returns (pred, loss, {key: grad}, {bn key: updated running stat}); sd tensors must be leaf tensors with requires_grad."""
    _TRAIN["on"], _TRAIN["stats"] = True, {}
    try:
        pred = hm_forward(left, right, sd)
    finally:
        _TRAIN["on"] = False
    n = pred.shape[1] // 2
    if plen is not None:
        sq = torch.sqrt(plen)[..., None, None]
        loss = lam * (F.mse_loss(pred[:, :n] / sq[:, :n], gt[:, :n] / sq[:, :n]) + F.mse_loss(pred[:, n:] / sq[:, n:], gt[:, n:] / sq[:, n:]))
    else:
        loss = lam * (F.mse_loss(pred[:, :n], gt[:, :n]) + F.mse_loss(pred[:, n:], gt[:, n:]))
    keys = [k for k, v in sd.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [sd[k] for k in keys], allow_unused=True)
    return pred.detach(), loss.detach(), dict(zip(keys, grads)), dict(_TRAIN["stats"])


def to_torch_sd(np_sd, dtype=torch.float32):
    out = {}
    for k, v in np_sd.items():
        t = torch.from_numpy(v)
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out
