"""CPU-side checks of the C ABI: the library loads, exports every symbol include/egotap.h declares,
and its host-side argument checking behaves (no kernel is launched here)."""
import ctypes as C
import os
import re

import pytest

from egotap_amd import lib as L

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="egotap.h"):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(egotap_[a-z0-9_]+)\s*\(", text)))


def test_header_and_library_agree():
    lib = L.load()
    boundary, hooks = _declared(), _declared("egotap_debug.h")
    assert len(boundary) >= 15
    # the drop-in boundary holds no test / measurement hook, the hook header nothing else
    assert not [n for n in boundary if "debug" in n or "timing" in n], "hooks belong in egotap_debug.h"
    assert hooks and all("debug" in n or "timing" in n for n in hooks) and not set(hooks) & set(boundary)
    names = sorted(boundary + hooks)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"
    assert sorted(L.exported_symbols()) == names
    # and the library exports nothing beyond the two headers
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", L._build.LIB], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r"\b(egotap_[a-z0-9_]+)$", out, flags=re.M)))
    assert exported == names, sorted(set(exported) ^ set(names))
    assert lib.egotap_abi_version() == 2
    # the library reads no environment variable
    src = "".join(open(os.path.join(REPO, "egotap_amd", "csrc", f)).read() for f in os.listdir(os.path.join(REPO, "egotap_amd", "csrc")))
    assert "getenv" not in src


def _cfg(**kw):
    base = dict(n_joints_hm=15, estimate_head=1, hm_size=64, hidden=128, vit_dim=1024, vit_heads=8, vit_layers=3,
                patch=16, pu_hidden=512)
    base.update(kw)
    return L.EgotapConfig(C.sizeof(L.EgotapConfig), *[base[k] for k in (
        "n_joints_hm", "estimate_head", "hm_size", "hidden", "vit_dim", "vit_heads", "vit_layers", "patch", "pu_hidden")])


def test_create_rejects_bad_config():
    lib = L.load()
    h = C.c_void_p()
    bad = _cfg(hm_size=60)
    assert lib.egotap_create(C.byref(bad), C.byref(h)) == 1
    assert b"multiple of 16" in lib.egotap_last_error()
    bad = _cfg()
    bad.struct_bytes = 4
    assert lib.egotap_create(C.byref(bad), C.byref(h)) == 1


def test_bind_checks_shape_and_reports_unbound():
    lib = L.load()
    h = C.c_void_p()
    cfg = _cfg()
    assert lib.egotap_create(C.byref(cfg), C.byref(h)) == 0
    n = C.c_int()
    assert lib.egotap_unbound_count(h, L.NET_LIFT, C.byref(n)) == 0
    assert n.value == 4 + 3 * 16 + 2 + 2 * 3 * 6 + 14 + 2 + 2    # forward-needed keys of the UE lifting head
    key = b"pos_heatmap_encoder.vit.layernorm.weight"
    fake = C.c_void_p(0x1000)
    assert lib.egotap_bind_param(h, L.NET_LIFT, key, fake, 1000, L.F32) == 1
    assert b"expected 1024" in lib.egotap_last_error()
    assert lib.egotap_bind_param(h, L.NET_LIFT, key, C.c_void_p(0x1004), 1024, L.F32) == 1   # misaligned
    assert lib.egotap_bind_param(h, L.NET_LIFT, key, fake, 1024, L.F32) == 0
    assert lib.egotap_unbound_count(h, L.NET_LIFT, C.byref(n)) == 0
    before = n.value
    # forward with unbound parameters must fail loudly, before any launch
    ws = C.c_size_t()
    assert lib.egotap_lift_workspace_bytes(h, 2, C.byref(ws)) == 0 and ws.value > 0
    rc = lib.egotap_lift_forward(h, C.c_void_p(0x1000), 2, C.c_void_p(0x2000), C.c_void_p(0x10000), ws.value, None)
    assert rc == 3 and b"not bound" in lib.egotap_last_error()
    assert before > 0
    lib.egotap_destroy(h)


def test_workspace_grows_with_batch():
    lib = L.load()
    h = C.c_void_p()
    cfg = _cfg()
    assert lib.egotap_create(C.byref(cfg), C.byref(h)) == 0
    a, b = C.c_size_t(), C.c_size_t()
    lib.egotap_lift_workspace_bytes(h, 1, C.byref(a))
    lib.egotap_lift_workspace_bytes(h, 256, C.byref(b))
    fixed = 64 << 20                      # split-K partial sums of the small-batch GEMMs: the same 64 MiB at every batch size
    assert a.value > fixed and 200 * (a.value - fixed) < b.value - fixed < 260 * (a.value - fixed)
    assert b.value < 8 << 30
    lib.egotap_destroy(h)


def test_module_mirror_has_reference_state_dict():
    import types
    import numpy as np
    from egotap_amd import networks, spec
    g = np.load(os.path.join(REPO, "tests", "golden", "lift_fwd_ue_b2.npz"))
    opt = types.SimpleNamespace(joint_preset="UnrealEgo", num_heatmap=15, num_rot_heatmap=15, heatmap_type="sin",
                                ae_hidden_size=128, patched_heatmap_ae=True, skel_layer="PU", load_size_heatmap=[64, 64])
    net = networks.EgoTAPAutoEncoder(opt, input_channel_scale=2)
    sd = net.state_dict()
    assert list(sd.keys()) == list(g["state_keys"])
    assert ["x".join(str(d) for d in v.shape) for v in sd.values()] == list(g["state_shapes"])
    assert [k for k, _ in net.named_parameters()] == list(g["param_keys"])
    net.eval()
    import torch
    with pytest.raises(L.EgotapError):
        net(torch.zeros(1, 90, 64, 64))          # CPU tensor: no fallback, must raise
    net.train()
    with pytest.raises(L.EgotapError):
        net(torch.zeros(1, 90, 64, 64))          # training mode: same rule, no CPU path


def test_estimator_alias_buffers_stay_tied_through_module_apply():
    """The reference registers the ResNet's children twice (backbone.backbone.backbone.* and backbone.backbone.layerK.*: the same
    nn.BatchNorm2d modules, net_architecture.py:68-73), so both key families are ONE tensor there and its load_state_dict reads the
    layerK.* keys last.  Module._apply (.to / .cuda / .double) replaces buffers per registration: the mirror re-ties them, so a
    train-mode forward that updates running statistics can never save stale aliases."""
    import types
    import torch
    from egotap_amd import networks
    opt = types.SimpleNamespace(joint_preset="UnrealEgo", num_heatmap=15, num_rot_heatmap=0, heatmap_type="none", ae_hidden_size=128,
                                load_size_heatmap=[64, 64])
    net = networks.HeatMap_UnrealEgo_Shared(opt, "resnet18", 2).double().float()
    sd = net.state_dict(keep_vars=True)
    pairs = [("backbone.backbone.backbone.bn1", "backbone.backbone.layer0.1"),
             ("backbone.backbone.backbone.layer3.1.bn2", "backbone.backbone.layer3.1.bn2"),
             ("backbone.backbone.backbone.layer2.0.downsample.1", "backbone.backbone.layer2.0.downsample.1")]
    for a, b in pairs:
        for leaf in ("running_mean", "running_var", "num_batches_tracked", "weight"):
            assert sd[f"{a}.{leaf}"] is sd[f"{b}.{leaf}"], (a, b, leaf)
    assert len(sd) == 258 and len(list(net.named_buffers())) == 60 and len(list(net.named_buffers(remove_duplicate=False))) == 120
    with torch.no_grad():
        sd["backbone.backbone.backbone.bn1.running_mean"].add_(1.0)
    assert torch.equal(net.state_dict()["backbone.backbone.layer0.1.running_mean"], net.state_dict()["backbone.backbone.backbone.bn1.running_mean"])


def test_host_side_argument_checks_of_the_newer_entry_points():
    """no GPU here: everything below must be rejected (or accepted) by host-side checks before any launch"""
    import torch
    from oracle import heatmap_synth_ref as R
    lib = L.load()
    h = C.c_void_p()
    L.check(lib.egotap_create(C.byref(_cfg()), C.byref(h)))
    try:
        for mode in (0, 1, 2):
            L.check(lib.egotap_set_precision(h, mode))
        assert lib.egotap_set_precision(h, 7) == 1 and b"unknown mode" in lib.egotap_last_error()
        assert lib.egotap_set_weight_scratch(h, C.c_void_p(8), 1024) == 1            # misaligned
        L.check(lib.egotap_set_weight_scratch(h, C.c_void_p(0), 0))                  # off
        assert lib.egotap_hmtrain_set_pack_buffer(h, C.c_void_p(4096), 16) == 1      # too small
        assert lib.egotap_hmtrain_pack_bytes() >= 1024 * 97 * 9 * 64
        # batch 0 is a no-op for the batched entry points, null pointers otherwise are errors
        L.check(lib.egotap_pose_metrics(None, None, 0, 16, None, None, None, None))
        assert lib.egotap_pose_metrics(None, None, 4, 16, None, None, None, None) == 1
        assert lib.egotap_attention(None, None, 1, 64, 1, 0, None) == 1
        assert lib.egotap_hmtrain_conv_wgrad(None, None, None, 1, 8, 8, 16, 3, 1, 0, 0, 0, 0, None, 0, None) == 1
        # [r5] the batch-statistics forward of a frozen estimator: workspace = whole-batch backbone maps + one decoder chunk (grows with both),
        # bf16 precision only, at least two frames, no launch on any refusal
        n1, n2, n3 = C.c_size_t(), C.c_size_t(), C.c_size_t()
        L.check(lib.egotap_hm_forward_bnbatch_workspace_bytes(h, 64, 16, C.byref(n1)))
        L.check(lib.egotap_hm_forward_bnbatch_workspace_bytes(h, 64, 64, C.byref(n2)))
        L.check(lib.egotap_hm_forward_bnbatch_workspace_bytes(h, 128, 16, C.byref(n3)))
        assert 0 < n1.value < n2.value and n1.value < n3.value
        assert n3.value - n1.value == 64 * (64 * 64 * 128 * 2) * (1 + 4 * (1 + 0.5 + 0.25 + 0.125))      # 8.5 MB of bf16 backbone maps per frame
        off, num = C.c_size_t(), C.c_int64()
        L.check(lib.egotap_hm_forward_bnbatch_intermediate(h, 64, 16, b"layer4", C.byref(off), C.byref(num)))
        assert num.value == 64 * 8 * 8 * 1024 and off.value % 256 == 0
        assert lib.egotap_hm_forward_bnbatch_intermediate(h, 64, 16, b"nope", C.byref(off), C.byref(num)) == 1
        L.check(lib.egotap_set_precision(h, 0))
        assert lib.egotap_hm_forward_bnbatch(h, 1, C.c_void_p(256), C.c_void_p(256), 4, C.c_void_p(256), 30 * 4096, 0, C.c_void_p(256), 1 << 40, None) == 1
        assert b"EGOTAP_PREC_BF16" in lib.egotap_last_error()
        L.check(lib.egotap_set_precision(h, 2))
        assert lib.egotap_hm_forward_bnbatch(h, 1, C.c_void_p(256), C.c_void_p(256), 1, C.c_void_p(256), 30 * 4096, 0, C.c_void_p(256), 1 << 40, None) == 1      # one frame
        L.check(lib.egotap_hm_forward_bnbatch(h, 1, None, None, 0, None, 0, 0, None, 0, None))                                                                     # batch 0: no-op
        assert lib.egotap_hm_forward_bnbatch(h, 0, C.c_void_p(256), C.c_void_p(256), 4, C.c_void_p(256), 30 * 4096, 0, C.c_void_p(256), 1 << 40, None) == 1      # the lifting head's id
        assert lib.egotap_hm_forward_bnbatch(h, 1, C.c_void_p(256), C.c_void_p(256), 4, C.c_void_p(256), 30 * 4096, 0, C.c_void_p(256), 1 << 40, None) == 3      # nothing bound
        # attention: any sequence that is a multiple of 4 from 32 on is a shape the fp32 kernel takes (null pointers are still refused first)
        assert lib.egotap_attention_f32(None, None, 1, 144, 8, None) == 1 and b"null" in lib.egotap_last_error()
        assert lib.egotap_attention_f32(C.c_void_p(256), C.c_void_p(256), 1, 30, 8, None) == 1 and b"at least 32" in lib.egotap_last_error()
        L.check(lib.egotap_set_precision(h, 0))
        # [r5] the measurement switch of the convolution operands' addressing (host state only): 0 / 1, anything else refused by name
        assert lib.egotap_debug_conv_addressing(2) == 1 and b"mode must be 0" in lib.egotap_last_error()
        assert lib.egotap_debug_conv_addressing(1) == 0 and lib.egotap_debug_conv_addressing(0) == 0
    finally:
        lib.egotap_destroy(h)
    # Python wrappers refuse CPU tensors (no CPU fallback) and inconsistent shapes
    with pytest.raises(L.EgotapError):
        L.synth_heatmaps(torch.zeros(1, 16, 2), torch.zeros(1, 16, 2), torch.zeros(1, 16, 3))
    with pytest.raises(L.EgotapError):
        L.pose_metrics(torch.zeros(2, 16, 3), torch.zeros(2, 16, 3))
    assert L.KINEMATIC_PARENTS == R.KINEMATIC_PARENTS and len(L.KINEMATIC_PARENTS["EgoCap"]) == 18
    assert sorted(L.PRECISIONS) == ["bf16", "bf16x3", "f32"]


def test_bottleneck_backbones_spec_and_module():
    """--model_name resnet50 / resnet101 (net_architecture.py:61-64, 108-111: torchvision Bottleneck ResNets, feature_scale 4): torchvision's
    key names and shapes (320 / 626 state_dict entries, 25.56 M / 44.55 M parameters), decoder widths x 4, the reference's aliases; fp32 eval
    only -- the bf16 modes and stage-1 training refuse them by name."""
    import math
    import torch
    from egotap_amd import networks, spec
    from egotap_amd.options import preset_defaults
    for name, n_entries, n_params in (("resnet50", 320, 25557032), ("resnet101", 626, 44549160)):
        rs = spec.resnet18_spec(name)
        assert len(rs) == n_entries
        assert sum(math.prod(shp) for k, shp in rs if "running" not in k and "num_batches" not in k) == n_params
    d = dict(spec.resnet18_spec("resnet50"))
    assert d["layer1.0.conv1.weight"] == (64, 64, 1, 1) and d["layer1.0.conv3.weight"] == (256, 64, 1, 1) and d["layer1.0.downsample.0.weight"] == (256, 64, 1, 1)
    assert d["layer2.0.conv1.weight"] == (128, 256, 1, 1) and d["layer2.0.conv2.weight"] == (128, 128, 3, 3) and d["layer4.2.conv3.weight"] == (2048, 512, 1, 1)
    assert "layer3.22.conv3.weight" in dict(spec.resnet18_spec("resnet101")) and d["fc.weight"] == (1000, 2048)
    hs = {k: shp for k, shp, _ in spec.hm_state_spec(15, "resnet50")}
    assert hs["after_backbone.conv_up3.0.weight"] == (4096, 6160, 3, 3) and hs["after_backbone.layer3_1x1.0.weight"] == (2064, 2048, 1, 1)
    assert hs["after_backbone.conv_up1.0.weight"] == (2048, 2560, 3, 3) and hs["after_backbone.conv_heatmap.weight"] == (30, 2048, 1, 1)
    # the module: keys in the reference's order, aliases are the same objects (built on the meta device: 1.7 GB of decoder weights otherwise)
    opt = preset_defaults("UnrealEgo")
    opt.num_rot_heatmap = 0
    with torch.device("meta"):
        net = networks.HeatMap_UnrealEgo_Shared(opt, "resnet50", input_channel_scale=2)
    assert list(net.state_dict().keys()) == [k for k, _, _ in spec.hm_state_spec(15, "resnet50")]
    sd = net.state_dict(keep_vars=True)
    assert sd["backbone.backbone.layer3.5.conv3.weight"] is sd["backbone.backbone.backbone.layer3.5.conv3.weight"]
    assert net.bottleneck and net.blocks == (3, 4, 6, 3)
    with pytest.raises(NotImplementedError, match="fp32 only"):
        net.set_precision("bf16")
    net.train()
    with pytest.raises(L.EgotapError):
        net(torch.zeros(1, 3, 256, 256), torch.zeros(1, 3, 256, 256))          # GPU only, as every net
    assert spec.hm_feature_scale("resnet101") == 4 and spec.hm_feature_scale("resnet34") == 1
    with pytest.raises(NotImplementedError):
        spec.hm_blocks("resnet50")                                               # the BasicBlock-only paths (one-call C forward, bf16, training)


def test_resnet34_backbone_spec_module_and_handle():
    """--model_name resnet34 (net_architecture.py:59-60, 104-105: torchvision resnet34, feature_scale 1): BasicBlocks (3, 4, 6, 3) per
    stage under torchvision's key names, the same decoder (the Bottleneck ResNets resnet50 / 101: test_bottleneck_backbones_spec_and_module -- fp32 evaluation only)."""
    import torch
    from egotap_amd import networks, spec
    from egotap_amd.options import preset_defaults
    rs = spec.resnet18_spec("resnet34")
    assert len(rs) == 218 and len(spec.resnet18_spec()) == 122                     # torchvision's state_dict sizes
    keys = [k for k, _ in rs]
    assert "layer3.5.conv2.weight" in keys and "layer3.6.conv1.weight" not in keys and "layer1.2.bn2.running_var" in keys
    assert [k for k in keys if "downsample.0" in k] == [f"layer{i}.0.downsample.0.weight" for i in (2, 3, 4)]
    assert dict(rs)["layer2.0.conv1.weight"] == (128, 64, 3, 3) and dict(rs)["layer2.3.conv1.weight"] == (128, 128, 3, 3)
    opt = preset_defaults("UnrealEgo")
    opt.num_rot_heatmap = 0
    net = networks.HeatMap_UnrealEgo_Shared(opt, "resnet34", input_channel_scale=2)
    want = [k for k, _, _ in spec.hm_state_spec(15, "resnet34")]
    assert list(net.state_dict().keys()) == want
    sd = net.state_dict(keep_vars=True)
    assert sd["backbone.backbone.layer3.5.conv2.weight"] is sd["backbone.backbone.backbone.layer3.5.conv2.weight"]      # the reference's aliases
    with pytest.raises(NotImplementedError):
        networks.HeatMap_UnrealEgo_Shared(opt, "resnet152", input_channel_scale=2)
    lib = L.load()
    counts = {}
    for name, blocks in (("resnet18", (0, 0, 0, 0)), ("resnet34", (3, 4, 6, 3))):
        cfg = _cfg()
        for i in range(4):
            cfg.hm_blocks[i] = blocks[i]
        h, n = C.c_void_p(), C.c_int()
        assert lib.egotap_create(C.byref(cfg), C.byref(h)) == 0
        assert lib.egotap_unbound_count(h, L.NET_HM_POS, C.byref(n)) == 0
        counts[name] = n.value
        lib.egotap_destroy(h)
    assert counts["resnet34"] - counts["resnet18"] == 8 * 2 * 5                    # eight more blocks: two convolutions + two BatchNorms (4 tensors) each
    bad = _cfg()
    bad.hm_blocks[2] = 7
    h = C.c_void_p()
    assert lib.egotap_create(C.byref(bad), C.byref(h)) == 1 and b"hm_blocks" in lib.egotap_last_error()


def test_weight_gradient_split_count_host_logic():
    """[r3] wgrad_pick_splits (csrc/gemm_tn_f32.h), the host-side choice behind every weight-gradient launch: covers the contraction exactly, stays within
    the workspace, fills whole rounds of the 256 CUs where round 2's rule left a partial third round, and answers the same question the same way."""
    lib = L.load()
    per = C.c_int()

    def pick(tiles, slabs, n, mem=256 << 20, cu=256, lds=118016, slab_us=7.68, fixed=4.0):
        s = lib.egotap_debug_wgrad_splits(tiles, slabs, n, mem, cu, lds, slab_us, fixed, C.byref(per))
        return s, per.value

    # conv_up1 of a 32-frame stage-1 step: 80 tiles, 32 images x 64 rows; round 2: 7 splits of whole images = 560 workgroups (three rounds for 2.2)
    s, p = pick(80, 32 * 64, 512 * 640 * 9)
    assert (s, p) == (16, 128) and (80 * s) % 256 == 0
    # a 64 -> 64 layer of layer1 with the 2 x 2 wave layout: one tile, 64 images x 64 rows -> every CU gets one workgroup
    s, p = pick(1, 64 * 64, 64 * 64 * 9, lds=136192)
    assert s == 256 and p == 16
    # the ViT's D x D weight gradients at B = 256 (4608 slabs of 32 rows): one full round
    s, p = pick(16, 4608, 1024 * 1024, mem=512 << 20, lds=133120, slab_us=3.41, fixed=6.0)
    assert (s, p) == (16, 288)
    for tiles, slabs, n in ((392, 512, 1024 * 1540 * 9), (3, 100, 4096), (7, 1, 1 << 20), (160, 1024, 512 * 1280 * 9)):
        s, p = pick(tiles, slabs, n)
        assert s >= 1 and p >= 1 and (s - 1) * p < slabs <= s * p              # every split is non-empty, together they cover the slabs
        assert s * n * 4 <= 256 << 20
        assert pick(tiles, slabs, n) == (s, p)                                 # (second call: the per-thread memo)
    assert pick(4, 64, 1 << 20, mem=(1 << 22) - 1)[0] == 0                      # not even one slab fits
    assert pick(4, 64, 1 << 20, mem=3 << 22)[0] <= 3                            # the workspace bounds the count
