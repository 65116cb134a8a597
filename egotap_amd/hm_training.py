"""Autograd glue of the heatmap estimator's training step (stage 1 of the reference: model/heatmap_shared_model.py:98-172 drives
HeatMap_UnrealEgo_Shared, model/net_architecture.py:25-173, in train mode).  PyTorch allocates tensors and connects the
gradient to the parameters; every operator is a HIP kernel behind the C ABI (egotap_hmtrain_*, hm_ops.py)."""
from __future__ import annotations

import torch

from . import hm_ops as H
from . import train_ops as T

STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))
BB = "backbone.backbone.backbone."
AB = "after_backbone."


def _eyes(t, B):
    """[2B, C, s, s] with image n = 2b + eye  ->  the two per-eye Views (channel slices of the [B, 2C, s, s] view)"""
    C_ = t.shape[1]
    v = t.view(B, 2 * C_, t.shape[2], t.shape[3])
    return [H.View(v, e * C_, C_) for e in range(2)]


def _bn_fwd(z, y, P, buf, k, B, res=None, relu=True):
    """BatchNorm2d in train mode, ONE EYE AT A TIME: the reference runs the shared backbone once on the left batch and once on
    the right batch (net_architecture.py:45-50), so batch statistics are per eye and the running stats move twice per step"""
    stats = []
    for e in range(2):
        stats.append(H.bn2d_fwd(_eyes(z, B)[e], _eyes(y, B)[e], P[k + ".weight"], P[k + ".bias"], buf[k + ".running_mean"], buf[k + ".running_var"],
                                res=_eyes(res, B)[e] if res is not None else None, relu=relu))
        buf[k + ".num_batches_tracked"] += 1
    return stats


def _bn_bwd(z, y, dy, P, k, stats, dz, G, B, dres=None, relu=True):
    for e in range(2):
        mean, rstd = stats[e]
        H.bn2d_bwd(_eyes(z, B)[e], _eyes(y, B)[e] if y is not None else None, _eyes(dy, B)[e], P[k + ".weight"], mean, rstd, _eyes(dz, B)[e],
                   G(k + ".weight"), G(k + ".bias"), dres=_eyes(dres, B)[e] if dres is not None else None, relu=relu, accumulate=e == 1)


def _param_items(net):
    """(key, Parameter) of every trainable tensor once (the backbone.backbone.layerK.* aliases are the same objects)"""
    return [(k, p) for k, p in net.named_parameters()]


def hm_arena_order(keys, n_stages=4):
    """Order of an estimator's trained tensors in its flat gradient arena = the order in which the backward FINISHES them, cut into
    buckets for the overlapped all-reduce (parallel.GradReducer): bucket 0 = the decoder (after_backbone.*: 2/3 of the
    parameters, complete before the backbone's backward starts), bucket 1 = layer4, bucket 2 = layer3 .. layer1 and the stem.
    Returns (keys in arena order, [bucket end index into that list])."""
    dec = [k for k in keys if k.startswith(AB)]
    l4 = [k for k in keys if k.startswith(f"{BB}layer{n_stages}.")]
    rest = [k for k in keys if k.startswith(BB) and k not in set(l4)]
    order = dec + l4 + rest
    assert sorted(order) == sorted(keys), "every trained tensor belongs to exactly one bucket"
    return order, [len(dec), len(dec) + len(l4), len(order)]


def _grad_arena(net, P):
    """views of the estimator's flat gradient arena, one per trainable tensor (allocated once per module and device); `bounds` = the
    bucket boundaries in floats"""
    dev = next(iter(P.values())).device
    ga = getattr(net, "_hm_grad_arena", None)
    sig = tuple((k, v.numel()) for k, v in P.items())
    if ga is None or ga["flat"].device != dev or ga["sig"] != sig:
        order, ends = hm_arena_order(list(P.keys()))
        offs, o = {}, 0
        for k in order:
            offs[k] = o
            o += (P[k].numel() + 63) // 64 * 64                  # 256-byte aligned slices
        bounds = [0] + [offs[order[e]] if e < len(order) else o for e in ends]
        ga = dict(flat=torch.zeros(o, dtype=torch.float32, device=dev), offs=offs, sig=sig, bounds=bounds)   # (zeros: the alignment gaps ride along in the all-reduce)
        net._hm_grad_arena = ga
    return {k: ga["flat"][ga["offs"][k]: ga["offs"][k] + v.numel()].view(v.shape) for k, v in P.items()}


def _reducer(net):
    """the overlapped gradient reducer of this estimator's training Function (a no-op for one rank)"""
    if getattr(net, "_grad_reducer", None) is None:
        from .parallel import GradReducer
        net._grad_reducer = GradReducer()
    return net._grad_reducer


class HmTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, left, right, *params):
        keys = [k for k, _ in _param_items(net)]
        P = dict(zip(keys, params))
        buf = dict(net.named_buffers())
        h = net._ensure_handle()
        dev = left.device
        left, right = left.detach().float().contiguous(), right.detach().float().contiguous()
        B, S0 = left.shape[0], left.shape[2]
        N2 = 2 * B
        new = lambda *shape: torch.empty(shape, device=dev)     # noqa: E731
        sv = {"B": B, "S0": S0}
        # frozen estimators of stage 2 (hm_train_forward_nograd): nothing is kept for a backward, every map is dropped as soon as its
        # consumer has run -- a 1024-frame batch peaks at a few concat buffers instead of every activation of the network
        keep = any(ctx.needs_input_grad[3:])
        with torch.no_grad():
            x0 = torch.stack([left, right], 1).reshape(N2, 3, S0, S0) if keep else None   # image n = 2b + eye (plumbing copy, needed by the stem wgrad)
            z0 = new(N2, 64, S0 // 2, S0 // 2)
            H.stem_fwd(left, right, P[BB + "conv1.weight"], z0)
            l0 = torch.empty_like(z0)
            m0 = _bn_fwd(z0, l0, P, buf, BB + "bn1", B)
            p0 = new(N2, 64, S0 // 4, S0 // 4)
            H.maxpool_fwd(l0, p0)
            if keep:
                sv.update(x0=x0, z0=z0, l0=l0, m0=m0, p0=p0)
            x, cin, side = p0, 64, S0 // 4
            del z0, l0, p0
            blocks, pyr = [], []
            for i, (c, st) in enumerate(STAGES, start=1):
                for b in range(net.blocks[i - 1]):
                    k = f"{BB}layer{i}.{b}."
                    stride = st if b == 0 else 1
                    so = side // stride
                    z1, y1 = new(N2, c, so, so), new(N2, c, so, so)
                    H.conv_fwd(h, x, P[k + "conv1.weight"], z1, taps=9, stride=stride)
                    m1 = _bn_fwd(z1, y1, P, buf, k + "bn1", B)
                    rec = dict(k=k, xin=x, stride=stride, cin=cin, c=c, z1=z1, y1=y1, m1=m1)
                    idt = x
                    if (k + "downsample.0.weight") in P:
                        zd, yd = new(N2, c, so, so), new(N2, c, so, so)
                        H.conv_fwd(h, x, P[k + "downsample.0.weight"], zd, taps=1, stride=stride)
                        md = _bn_fwd(zd, yd, P, buf, k + "downsample.1", B, relu=False)
                        rec.update(zd=zd, md=md)
                        idt = yd
                    z2, y2 = new(N2, c, so, so), new(N2, c, so, so)
                    H.conv_fwd(h, y1, P[k + "conv2.weight"], z2, taps=9, stride=1)
                    m2 = _bn_fwd(z2, y2, P, buf, k + "bn2", B, res=idt)
                    rec.update(z2=z2, y2=y2, m2=m2, level=i - 1 if b == net.blocks[i - 1] - 1 else None)
                    if keep:
                        blocks.append(rec)
                    x, cin, side = y2, c, so
                    del rec, z1, y1, z2, y2, idt
                    zd = yd = None
                pyr.append(x)
            # decoder on the channel-concatenated pyramids: [2B, C, s, s] viewed as [B, 2C, s, s]
            L = [t.view(B, 2 * t.shape[1], t.shape[2], t.shape[3]) for t in pyr]
            s64, s32, s16, s8 = L[0].shape[2], L[1].shape[2], L[2].shape[2], L[3].shape[2]
            u4 = new(B, 1024, s8, s8)
            H.conv_fwd(h, L[3], P[AB + "layer4_1x1.0.weight"], u4, bias=P[AB + "layer4_1x1.0.bias"], taps=1, relu=True)
            cat3 = new(B, 1540, s16, s16)
            H.upsample_fwd(u4, H.View(cat3, 0, 1024))
            H.conv_fwd(h, L[2], P[AB + "layer3_1x1.0.weight"], H.View(cat3, 1024, 516), bias=P[AB + "layer3_1x1.0.bias"], taps=1, relu=True)
            x3 = new(B, 1024, s16, s16)
            H.conv_fwd(h, cat3, P[AB + "conv_up3.0.weight"], x3, bias=P[AB + "conv_up3.0.bias"], taps=9, relu=True)
            cat2 = new(B, 1280, s32, s32)
            H.upsample_fwd(x3, H.View(cat2, 0, 1024))
            H.conv_fwd(h, L[1], P[AB + "layer2_1x1.0.weight"], H.View(cat2, 1024, 256), bias=P[AB + "layer2_1x1.0.bias"], taps=1, relu=True)
            x2 = new(B, 512, s32, s32)
            H.conv_fwd(h, cat2, P[AB + "conv_up2.0.weight"], x2, bias=P[AB + "conv_up2.0.bias"], taps=9, relu=True)
            cat1 = new(B, 640, s64, s64)
            H.upsample_fwd(x2, H.View(cat1, 0, 512))
            H.conv_fwd(h, L[0], P[AB + "layer1_1x1.0.weight"], H.View(cat1, 512, 128), bias=P[AB + "layer1_1x1.0.bias"], taps=1, relu=True)
            x1 = new(B, 512, s64, s64)
            H.conv_fwd(h, cat1, P[AB + "conv_up1.0.weight"], x1, bias=P[AB + "conv_up1.0.bias"], taps=9, relu=True)
            n_out = P[AB + "conv_heatmap.weight"].shape[0]
            out = new(B, n_out, s64, s64)
            H.conv_fwd(h, x1, P[AB + "conv_heatmap.weight"], out, bias=P[AB + "conv_heatmap.bias"], taps=1)
            if keep:
                sv.update(blocks=blocks, L=L, u4=u4, cat3=cat3, x3=x3, cat2=cat2, x2=x2, cat1=cat1, x1=x1)
        ctx.sv, ctx.net, ctx.keys, ctx.P = (sv if keep else None), net, keys, P
        return out

    @staticmethod
    def backward(ctx, dout):
        sv, net, keys, P = ctx.sv, ctx.net, ctx.keys, ctx.P
        h = net._ensure_handle()
        prec = getattr(net, "precision", "f32")
        B = sv["B"]
        N2 = 2 * B
        dev = dout.device
        # [r3] the gradients are views of ONE flat arena kept on the module and handed to the parameters directly (.grad), as the lifting head's
        # backward does: autograd's accumulation copies every tensor it cannot steal, and the optimizer updates a group whose gradients share an
        # allocation in one launch (EgotapAdamW / egotap_train_adamw_multi) instead of 76 launches.
        from .training import _held_grads, _publish_grads
        GA = _grad_arena(net, P)
        held = _held_grads(P, GA)
        ga = net._hm_grad_arena
        red = _reducer(net)                 # data parallel: the all-reduce of a bucket starts as soon as its kernels are enqueued
        red.begin(ga["flat"])
        G = {}
        dout = dout.detach().float().contiguous()

        def grad_of(key):
            if key not in G:
                G[key] = GA[key]
            return G[key]

        def bias_conv_bwd(name, dy_view, x_view, taps, want_dx=True, dx=None):
            """y = conv(x, w) + b: dW, db, (dX) from dZ"""
            w = P[AB + name + ".weight"]
            ks = 3 if taps == 9 else 1
            H.conv_wgrad(dy_view, x_view, grad_of(AB + name + ".weight"), ks=ks, stride=1, precision=prec)
            H.chansum(dy_view, grad_of(AB + name + ".bias"))
            if want_dx:
                H.conv_dgrad(h, dy_view, w, dx, taps=taps, stride=1)

        with torch.no_grad():
            L = sv["L"]
            s64, s32, s16, s8 = L[0].shape[2], L[1].shape[2], L[2].shape[2], L[3].shape[2]
            new = lambda *shape: torch.empty(shape, device=dev)    # noqa: E731
            # conv_heatmap
            dx1 = new(*sv["x1"].shape)
            bias_conv_bwd("conv_heatmap", dout, sv["x1"], 1, dx=dx1)
            # conv_up1 (relu) <- cat1
            dz = torch.empty_like(dx1)
            H.relu_bwd(sv["x1"], dx1, dz)
            dcat1 = new(*sv["cat1"].shape)
            bias_conv_bwd("conv_up1.0", dz, sv["cat1"], 9, dx=dcat1)
            dL = [None] * 4                 # gradients of the pyramid levels from the decoder, [B, 2C, s, s] views

            def skip_bwd(name, cat, dcat, c0, cn, level):
                """cat[:, c0:c0+cn] = relu(conv1x1(L[level]) + b)"""
                dzs = new(B, cn, cat.shape[2], cat.shape[3])
                H.relu_bwd(H.View(cat, c0, cn), H.View(dcat, c0, cn), dzs)
                dl = torch.empty_like(L[level])
                bias_conv_bwd(name, dzs, L[level], 1, dx=dl)
                dL[level] = dl

            skip_bwd("layer1_1x1.0", sv["cat1"], dcat1, 512, 128, 0)
            dx2 = new(*sv["x2"].shape)
            H.upsample_bwd(H.View(dcat1, 0, 512), dx2)
            dz = torch.empty_like(dx2)
            H.relu_bwd(sv["x2"], dx2, dz)
            dcat2 = new(*sv["cat2"].shape)
            bias_conv_bwd("conv_up2.0", dz, sv["cat2"], 9, dx=dcat2)
            skip_bwd("layer2_1x1.0", sv["cat2"], dcat2, 1024, 256, 1)
            dx3 = new(*sv["x3"].shape)
            H.upsample_bwd(H.View(dcat2, 0, 1024), dx3)
            dz = torch.empty_like(dx3)
            H.relu_bwd(sv["x3"], dx3, dz)
            dcat3 = new(*sv["cat3"].shape)
            bias_conv_bwd("conv_up3.0", dz, sv["cat3"], 9, dx=dcat3)
            skip_bwd("layer3_1x1.0", sv["cat3"], dcat3, 1024, 516, 2)
            du4 = new(*sv["u4"].shape)
            H.upsample_bwd(H.View(dcat3, 0, 1024), du4)
            dz = torch.empty_like(du4)
            H.relu_bwd(sv["u4"], du4, dz)
            dl4 = torch.empty_like(L[3])
            bias_conv_bwd("layer4_1x1.0", dz, L[3], 1, dx=dl4)
            dL[3] = dl4
            red.bucket_ready(ga["bounds"][0], ga["bounds"][1])          # the decoder's gradients are final
            # backbone, last block first.  dy = gradient of the current block's output (pyramid levels add their decoder share)
            blocks = sv["blocks"]
            dy = None
            last_stage = f"{BB}layer{len(STAGES)}."
            for bi in range(len(blocks) - 1, -1, -1):
                r = blocks[bi]
                k, c, cin, stride = r["k"], r["c"], r["cin"], r["stride"]
                if not k.startswith(last_stage) and blocks[bi + 1]["k"].startswith(last_stage):
                    red.bucket_ready(ga["bounds"][1], ga["bounds"][2])  # layer4 is done
                if r["level"] is not None:            # output of a stage = a pyramid level
                    share = dL[r["level"]].view(N2, c, r["y2"].shape[2], r["y2"].shape[3])
                    if dy is None:
                        dy = share
                    else:
                        T.add_inplace(dy, share)
                dz2, dres = torch.empty_like(r["z2"]), torch.empty_like(r["z2"])
                _bn_bwd(r["z2"], r["y2"], dy, P, k + "bn2", r["m2"], dz2, grad_of, B, dres=dres)
                H.conv_wgrad(dz2, r["y1"], grad_of(k + "conv2.weight"), ks=3, stride=1, precision=prec)
                dy1 = torch.empty_like(r["y1"])
                H.conv_dgrad(h, dz2, P[k + "conv2.weight"], dy1, taps=9, stride=1)
                dz1 = torch.empty_like(r["z1"])
                _bn_bwd(r["z1"], r["y1"], dy1, P, k + "bn1", r["m1"], dz1, grad_of, B)
                H.conv_wgrad(dz1, r["xin"], grad_of(k + "conv1.weight"), ks=3, stride=stride, precision=prec)
                dxin = torch.empty_like(r["xin"])
                if "zd" in r:
                    dzd = torch.empty_like(r["zd"])
                    _bn_bwd(r["zd"], None, dres, P, k + "downsample.1", r["md"], dzd, grad_of, B, relu=False)
                    H.conv_wgrad(dzd, r["xin"], grad_of(k + "downsample.0.weight"), ks=1, stride=stride)
                    H.conv_dgrad(h, dzd, P[k + "downsample.0.weight"], dxin, taps=1, stride=stride)
                    H.conv_dgrad(h, dz1, P[k + "conv1.weight"], dxin, taps=9, stride=stride, accumulate=True)
                else:
                    dxin.copy_(dres)                  # identity branch (plumbing copy), then the conv branch on top
                    H.conv_dgrad(h, dz1, P[k + "conv1.weight"], dxin, taps=9, stride=stride, accumulate=True)
                dy = dxin
            # stem
            dl0 = torch.empty_like(sv["l0"])
            H.maxpool_bwd(sv["l0"], dy, dl0)
            dz0 = torch.empty_like(sv["z0"])
            _bn_bwd(sv["z0"], sv["l0"], dl0, P, BB + "bn1", sv["m0"], dz0, grad_of, B)
            H.conv_wgrad(dz0, sv["x0"], grad_of(BB + "conv1.weight"), ks=7, stride=2)
            red.bucket_ready(ga["bounds"][2], ga["bounds"][3])
            red.finish()                    # the compute stream waits for the outstanding buckets; gradients arrive averaged
        ctx.sv = None
        _publish_grads({k: P[k] for k in G}, GA, held)
        return (None, None, None) + (None,) * len(keys)


def hm_train_forward(net, left, right):
    """differentiable train-mode forward of HeatMap_UnrealEgo_Shared: [B,3,S0,S0] x 2 -> [B, 2n, S0/4, S0/4]"""
    params = [p for _, p in _param_items(net)]
    net._bind(left.device)
    if getattr(net, "precision", "f32") != "f32":          # bf16 modes: scratch for the repacked conv weights (kept on the module)
        from . import lib as _lib
        import ctypes as C
        if getattr(net, "_pack", None) is None or net._pack.device != left.device:
            net._pack = torch.empty(_lib.load().egotap_hmtrain_pack_bytes(), dtype=torch.uint8, device=left.device)
        _lib.check(_lib.load().egotap_hmtrain_set_pack_buffer(net._ensure_handle(), C.c_void_p(net._pack.data_ptr()), net._pack.numel()))
    return HmTrainFn.apply(net, left, right, *params)


def hm_train_forward_nograd(net, left, right):
    """train-mode forward (batch-statistics BatchNorm2d per eye, running stats updated) with no graph: what the reference's FROZEN
    estimators compute while the lifting head trains under train.py:91 model.train() (egotap_autoencoder_model.py:179)"""
    was = net.training
    net.train()
    try:
        with torch.no_grad():
            return hm_train_forward(net, left, right)
    finally:
        net.train(was)
