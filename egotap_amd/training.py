"""Training step of the lifting head on the HIP operators: PyTorch is autograd glue only.

``LiftTrainFn`` is ONE torch.autograd.Function for the whole head (heatmaps + parameters -> pose): its forward runs
the training-mode network through libegotap_hip.so keeping the activations, its backward walks the layers in reverse
through the backward operators and hands every parameter gradient back to autograd, so ``loss.backward()`` /
``optimizer.step()`` in the wrapper read exactly like the reference (model/egotap_autoencoder_model.py:299-323).
``PoseLossFn`` is the loss (utils/loss.py:54-85), ``EgotapAdamW`` the optimizer (model/network.py:72-78) -- both HIP.

``LiftTrainFn`` runs the fp32 / bf16x3 arithmetic on fp32 tensors; ``LiftTrainBf16Fn`` is the reduced-precision mode (bf16
activations in HBM, fp32 master weights and accumulation) behind the wrapper's --use_amp.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as _lib
from . import spec as _spec
from . import train_ops as T


def _param_order(p):
    return [k for k, _ in _spec.lift_state_spec(p) if not _spec.is_buffer(k) and k not in _spec.LIFT_DEAD_KEYS]


def _arena_layout(p):
    """Order of the trained tensors in the flat gradient arena = the order in which the backward FINISHES them, cut into buckets:
    bucket 0 (head, propagation units, both FC encoders, the final LayerNorm, the last layer's output.dense.bias) is complete after
    the final LayerNorm's backward; bucket 1 + j (ViT layer L-1-j, plus the output.dense.bias of the layer below, whose gradient comes
    out of this layer's first LayerNorm backward) after that layer; the last bucket (patch embedding) at the end.
    Returns (keys in arena order, [bucket end index into keys])."""
    keys = _param_order(p)
    v = "pos_heatmap_encoder.vit."
    L = p.vit_layers
    layer_of = {}
    for k in keys:
        if k.startswith(v + "encoder.layer."):
            i = int(k[len(v + "encoder.layer."):].split(".")[0])
            layer_of[k] = i + 1 if k.endswith("output.dense.bias") and ".attention." not in k else i
    order, ends = [], []
    order += [k for k in keys if not k.startswith(v)] + [v + "layernorm.weight", v + "layernorm.bias"]
    order += [k for k in keys if layer_of.get(k) == L]
    ends.append(len(order))
    for i in reversed(range(L)):
        order += [k for k in keys if layer_of.get(k) == i]
        ends.append(len(order))
    order += [k for k in keys if k.startswith(v + "embeddings.")]
    ends.append(len(order))
    assert sorted(order) == sorted(keys) and len(set(order)) == len(order)
    return order, ends


def _grad_arena(net, P):
    """views of the flat gradient arena, one per trained tensor (allocated once per module and device)"""
    dev = next(iter(P.values())).device
    ga = getattr(net, "_grad_arena", None)
    if ga is None or ga["flat"].device != dev:
        order, ends = _arena_layout(net.preset)
        offs, o = {}, 0
        for k in order:
            offs[k] = o
            o += (P[k].numel() + 63) // 64 * 64                  # 256-byte aligned slices
        flat = torch.empty(o, dtype=torch.float32, device=dev)
        bounds = [0] + [offs[order[e]] if e < len(order) else o for e in ends]
        ga = dict(flat=flat, offs=offs, order=order, bounds=bounds)
        net._grad_arena = ga
    G = {k: ga["flat"][ga["offs"][k]: ga["offs"][k] + P[k].numel()].view(P[k].shape) for k in P}
    return ga, G


def _held_grads(P, G):
    """A backward writes the whole arena (accumulate = 0 in every gradient kernel).  Parameters whose .grad still IS their arena
    view (a second backward before zero_grad: gradient accumulation, several losses) would lose the gradient they hold, so it is
    copied aside here, before the kernels run, and added back by _publish_grads.  Empty in the reference's loop (zero_grad, one
    backward, step: model/egotap_autoencoder_model.py:299-323)."""
    return {k: prm.grad.clone() for k, prm in P.items() if prm.grad is not None and prm.grad.data_ptr() == G[k].data_ptr()}


def _publish_grads(P, G, held=None):
    """hand the arena views to the parameters directly (.grad), instead of returning them through autograd's accumulation (which
    would copy 384 MB unless it can steal the buffers): the optimizer and the gradient reducer then work in place on the arena.
    A parameter that already holds a gradient (a second backward before zero_grad) accumulates: into its own tensor when that is
    not the arena view, through the copy _held_grads took when it is."""
    for k, prm in P.items():
        if prm.grad is None:
            prm.grad = G[k]
        elif prm.grad.data_ptr() != G[k].data_ptr():
            prm.grad.add_(G[k])
        elif held is not None and k in held:
            G[k].add_(held[k])
        else:
            raise RuntimeError(f"{k}: .grad aliases the gradient arena but was not saved before the backward overwrote it")


_train_ws = T.Scratch()


def release_scratch(net=None):
    """drop the grow-only scratch buffers of the training path (between benchmark legs; the next step allocates them again)"""
    _train_ws.buf = None
    T._scratch.buf = None
    if net is not None:
        net._saved_pool = None


def _bind_grads(net, h, ga, G):
    """egotap_bind_grad for every trained tensor, once per arena"""
    sig = ga["flat"].data_ptr()
    if getattr(net, "_grads_bound", None) != sig:
        lib = _lib.load()
        for k, g in G.items():
            _lib.check(lib.egotap_bind_grad(h, k.encode(), C.c_void_p(g.data_ptr()), g.numel()))
        net._grads_bound = sig


class LiftTrainOneCallFn(torch.autograd.Function):
    """The training step (fp32 / bf16x3 arithmetic on fp32 tensors, or the bf16-storage step under EGOTAP_PREC_BF16) on the one-call ABI: egotap_lift_forward_train keeps the activations in one caller-owned buffer,
    egotap_lift_backward writes every gradient into the flat arena (bound once with egotap_bind_grad) and records one event per arena
    bucket, behind which the bucket's all-reduce starts on a side stream while the rest of the backward runs (parallel.GradReducer)."""

    @staticmethod
    def forward(ctx, net, hm, *params):
        p = net.preset
        keys = _param_order(p)
        P = dict(zip(keys, params))
        dev = hm.device
        h = net._ensure_handle()
        net._bind(dev)
        B = hm.shape[0]
        net._act_scratch(B, dev)
        lib = _lib.load()
        hm = hm.detach().float().contiguous()
        sb, wb = C.c_size_t(), C.c_size_t()
        _lib.check(lib.egotap_lift_train_bytes(h, B, C.byref(sb), C.byref(wb)))
        # the activations' buffer is kept by the module between steps (no 30 GB allocation per step, whose cost depends on what else
        # the process has cached); a second forward before the first one's backward gets a buffer of its own
        pool = getattr(net, "_saved_pool", None)
        if pool is not None and not pool["busy"] and pool["buf"].numel() >= sb.value and pool["buf"].device == dev:
            saved = pool["buf"]
            pool["busy"] = True
        else:
            # too small, another device, or still marked busy.  Busy means a forward whose backward has not run: either it is
            # still to come (two forwards, then two backwards: its ctx holds the buffer, dropping the module's reference frees
            # nothing early) or it never will (a no_grad train-mode forward, an exception): then this reference was the last one
            # and releasing it BEFORE the new allocation keeps the peak at one buffer (~30 GB at B = 1024) instead of two.
            net._saved_pool = pool = None
            saved = torch.empty(sb.value, dtype=torch.uint8, device=dev)
            net._saved_pool = pool = dict(buf=saved, busy=True)
        ws = _train_ws.get(wb.value, dev)
        pose = torch.empty((B, p.out_joints, 3), dtype=torch.float32, device=dev)
        _lib.check(lib.egotap_lift_forward_train(h, T._p(hm), B, T._p(pose), T._p(saved), saved.numel(), T._p(ws), ws.numel(), T._s()))
        for k, b in net.named_buffers():
            if k.endswith("num_batches_tracked"):
                b.add_(1)
        ctx.egotap = dict(net=net, P=P, keys=keys, B=B, hm=hm, saved=saved, wb=wb.value, pool=pool)
        return pose

    @staticmethod
    def backward(ctx, dpose):
        S = ctx.egotap
        net, P, keys, B = S["net"], S["P"], S["keys"], S["B"]
        h = net._ensure_handle()
        lib = _lib.load()
        dev = dpose.device
        ga, G = _grad_arena(net, P)
        held = _held_grads(P, G)
        _bind_grads(net, h, ga, G)
        red = net._reducer()
        red.begin(ga["flat"])
        nb = len(ga["bounds"]) - 1
        events, evp = [], None
        if red.active():
            events = red.events(nb)
            evp = (C.c_void_p * nb)(*[e.cuda_event for e in events])
        ws = _train_ws.get(S["wb"], dev)
        dpose = dpose.detach().float().contiguous()
        _lib.check(lib.egotap_lift_backward(h, T._p(S["hm"]), T._p(dpose), B, T._p(S["saved"]), S["saved"].numel(), T._p(ws), ws.numel(),
                                            evp, len(events), T._s()))
        for k, e in enumerate(events):
            red.bucket_ready(ga["bounds"][k], ga["bounds"][k + 1], after=e)
        red.finish()
        _publish_grads(P, G, held)
        S["pool"]["busy"] = False
        ctx.egotap = None
        return (None, None) + (None,) * len(keys)


class LiftTrainFn(torch.autograd.Function):
    """The same step composed operator by operator from Python (what the one-call ABI does inside the library): kept as the
    reference composition for tests/test_gpu_train_step.py (bit-identical gradients) and for `net.one_call_training = False`."""

    @staticmethod
    def forward(ctx, net, hm, *params):
        p = net.preset
        keys = _param_order(p)
        P = dict(zip(keys, params))
        dev = hm.device
        h = net._ensure_handle()
        # attention: exact fp32 unless the whole step is bf16.  Recomputing P = exp(S - lse) from split-bf16 scores (2^-16 per
        # product) costs ~1e-4 relative in every probability, and gradients that are sums with heavy cancellation (mask_token)
        # then miss the reference-golden gate (2.6 % of the tensor's typical magnitude against 0.5 %): bf16x3 stays fp32-grade.
        prec = "bf16" if getattr(net, "precision", "f32") == "bf16" else "f32"
        net._bind(dev)
        net._act_scratch(hm.shape[0], dev)
        B, D, seq, heads, J, T_, hid = hm.shape[0], p.vit_dim, p.seq, p.vit_heads, p.n_joints_hm, p.tokens, p.hidden
        M, BT = B * seq, B * T_
        lib = _lib.load()
        st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
        v = "pos_heatmap_encoder.vit."
        hm = hm.detach().float().contiguous()
        saved = {"hm": hm}
        x = torch.empty((M, D), dtype=torch.float32, device=dev)
        _lib.check(lib.egotap_train_patch_fwd(h, T._p(hm), B, T._p(P[v + "embeddings.patch_embeddings.projection.weight"]),
                                              T._p(P[v + "embeddings.patch_embeddings.projection.bias"]), T._p(P[v + "embeddings.mask_token"]),
                                              T._p(P[v + "embeddings.position_embeddings"]), T._p(x), st()))
        layers = []
        for i in range(p.vit_layers):
            l = f"{v}encoder.layer.{i}."
            a = l + "attention.attention."
            y1, m1, r1 = T.layernorm_fwd(x, P[l + "layernorm_before.weight"], P[l + "layernorm_before.bias"])
            qkv = torch.empty((M, 3 * D), dtype=torch.float32, device=dev)
            _lib.check(lib.egotap_train_qkv_fwd(h, T._p(y1), T._p(P[a + "query.weight"]), T._p(P[a + "query.bias"]), T._p(P[a + "key.weight"]),
                                                T._p(P[a + "key.bias"]), T._p(P[a + "value.weight"]), T._p(P[a + "value.bias"]), T._p(qkv), M, D, st()))
            ctx_, lse = T.attention_fwd(qkv, B, seq, heads, prec)
            xm = T.gemm_nt(h, ctx_, P[l + "attention.output.dense.weight"], P[l + "attention.output.dense.bias"], M, D, D, epi=T.TE_BIAS_RES, r=x)
            y2, m2, r2 = T.layernorm_fwd(xm, P[l + "layernorm_after.weight"], P[l + "layernorm_after.bias"])
            z = torch.empty((M, 4 * D), dtype=torch.float32, device=dev)
            hid_ = T.gemm_nt(h, y2, P[l + "intermediate.dense.weight"], P[l + "intermediate.dense.bias"], M, 4 * D, D, epi=T.TE_BIAS_GELU_SAVE, z=z)
            xo = T.gemm_nt(h, hid_, P[l + "output.dense.weight"], P[l + "output.dense.bias"], M, D, 4 * D, epi=T.TE_BIAS_RES, r=xm)
            layers.append(dict(x=x, m1=m1, r1=r1, y1=y1, qkv=qkv, ctx=ctx_, lse=lse, xm=xm, m2=m2, r2=r2, y2=y2, z=z, hid=hid_))
            x = xo
        tokens, mf, rf = T.layernorm_fwd(x, P[v + "layernorm.weight"], P[v + "layernorm.bias"])
        saved.update(layers=layers, xf=x, mf=mf, rf=rf, tokens=tokens)

        def encoder(name, loader, src, k1):
            acts = []
            a_in, ld, K = src, loader, k1
            for j, n_out in enumerate((2048, 512, hid), start=1):
                f = f"{name}.fc{j}"
                zz = T.gemm_nt(h, a_in, P[f + ".fc.weight"], P[f + ".fc.bias"], BT, n_out, K, loader=ld, epi=T.TE_BIAS)
                bufs = dict(net.named_buffers())
                yy, mean, rstd = T.bn_lrelu_fwd(zz, P[f + ".bn.weight"], P[f + ".bn.bias"], bufs[f + ".bn.running_mean"], bufs[f + ".bn.running_var"])
                bufs[f + ".bn.num_batches_tracked"].add_(1)
                acts.append(dict(a_in=a_in, z=zz, y=yy, mean=mean, rstd=rstd, loader=ld, K=K, N=n_out, name=f))
                a_in, ld, K = yy, T.LD_PLAIN, n_out
            return acts

        pos_acts = encoder("pos_heatmap_encoder", T.LD_TOKENS, tokens, p.ppd * p.ppd * D)
        rot_acts = encoder("rot_heatmap_encoder", T.LD_ROT, hm, 2 * p.hm_size * p.hm_size)
        posz, rotz = pos_acts[-1]["y"], rot_acts[-1]["y"]
        nb, off = C.c_size_t(), C.c_size_t()
        _lib.check(lib.egotap_train_pu_saved_bytes(h, B, C.byref(nb), C.byref(off)))
        pu_saved = torch.empty(nb.value, dtype=torch.uint8, device=dev)
        _lib.check(lib.egotap_train_pu_fwd(h, T._p(posz), T._p(rotz), B, T._p(pu_saved), pu_saved.numel(), st()))
        hs1 = pu_saved[off.value: off.value + 4 * J * B * p.pu_hidden].view(torch.float32)
        pose = torch.empty((B, p.out_joints, 3), dtype=torch.float32, device=dev)
        _lib.check(lib.egotap_train_pose_head_fwd(h, T._p(posz), T._p(hs1), B, T._p(pose), st()))
        saved.update(pos_acts=pos_acts, rot_acts=rot_acts, pu_saved=pu_saved, hs1=hs1, P=P, keys=keys, net=net, B=B)
        ctx.egotap = saved
        return pose

    @staticmethod
    def backward(ctx, dpose):
        S = ctx.egotap
        net, P, keys, B = S["net"], S["P"], S["keys"], S["B"]
        p = net.preset
        h = net._ensure_handle()
        # attention: exact fp32 unless the whole step is bf16.  Recomputing P = exp(S - lse) from split-bf16 scores (2^-16 per
        # product) costs ~1e-4 relative in every probability, and gradients that are sums with heavy cancellation (mask_token)
        # then miss the reference-golden gate (2.6 % of the tensor's typical magnitude against 0.5 %): bf16x3 stays fp32-grade.
        prec = "bf16" if getattr(net, "precision", "f32") == "bf16" else "f32"
        dev = dpose.device
        D, seq, heads, J, T_, hid, H = p.vit_dim, p.seq, p.vit_heads, p.n_joints_hm, p.tokens, p.hidden, p.pu_hidden
        M, BT = B * seq, B * T_
        lib = _lib.load()
        st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ga, G = _grad_arena(net, P)                            # every entry is fully overwritten below
        held = _held_grads(P, G)
        red = net._reducer()
        red.begin(ga["flat"])
        dpose = dpose.detach().float().contiguous()
        v = "pos_heatmap_encoder.vit."
        posz, rotz, hs1 = S["pos_acts"][-1]["y"], S["rot_acts"][-1]["y"], S["hs1"]
        # pose head + propagation units
        dposz, drotz = torch.empty_like(posz), torch.empty_like(rotz)
        dhs1 = torch.empty(J * B * H, dtype=torch.float32, device=dev)
        gw = G.get("global_mlp.pose_fcs.0.weight")
        gb = G.get("global_mlp.pose_fcs.0.bias")
        _lib.check(lib.egotap_train_pose_head_bwd(h, T._p(posz), T._p(hs1), T._p(dpose), B, T._p(dposz), T._p(dhs1),
                                                  T._p(G["pose_mlp.pose_fcs.0.weight"]), T._p(G["pose_mlp.pose_fcs.0.bias"]), T._p(gw), T._p(gb), 0, st()))
        wsb = C.c_size_t()
        _lib.check(lib.egotap_train_pu_bwd_ws_bytes(h, B, C.byref(wsb)))
        ws = torch.empty(wsb.value, dtype=torch.uint8, device=dev)
        ptrs = (C.c_void_p * 14)()
        c = "skel_sequential_layer.lstm_custom.layers."
        for i, nme in enumerate(("0.x2f", "0.x2h", "0.b2h", "0.h2h", "1.x2f", "1.x2h", "1.h2h")):
            ptrs[2 * i] = G[c + nme + ".weight"].data_ptr()
            ptrs[2 * i + 1] = G[c + nme + ".bias"].data_ptr()
        _lib.check(lib.egotap_train_pu_bwd(h, T._p(posz), T._p(rotz), B, T._p(S["pu_saved"]), T._p(dhs1), T._p(dposz), T._p(drotz), ptrs, 0,
                                           T._p(ws), ws.numel(), st()))
        del ws

        def encoder_bwd(acts, dy):
            """returns the gradient w.r.t. the gathered fc1 input rows (None for the rotation encoder: its input is data)"""
            for j in (2, 1, 0):
                a = acts[j]
                f = a["name"]
                dz = T.bn_lrelu_bwd(a["z"], a["y"], dy, P[f + ".bn.weight"], a["mean"], a["rstd"], G[f + ".bn.weight"], G[f + ".bn.bias"])
                T.gemm_tn(h, dz, a["a_in"], G[f + ".fc.weight"], BT, a["N"], a["K"], loader=a["loader"])
                T.colsum(dz, G[f + ".fc.bias"], BT, a["N"])
                if j == 0 and a["loader"] == T.LD_ROT:
                    return None
                wt = T.transpose(P[f + ".fc.weight"])                     # [K, N]
                dy = T.gemm_nt(h, dz, wt, None, BT, a["K"], a["N"], epi=T.TE_NONE)
            return dy

        encoder_bwd(S["rot_acts"], drotz)
        dA = encoder_bwd(S["pos_acts"], dposz)                            # [B*T, ppd*ppd*D], heatmap-major
        dtok = torch.empty((M, D), dtype=torch.float32, device=dev)
        _lib.check(lib.egotap_train_tokens_scatter(h, T._p(dA), T._p(dtok), B, st()))
        del dA
        dx = T.layernorm_bwd(S["xf"], dtok, P[v + "layernorm.weight"], S["mf"], S["rf"], G[v + "layernorm.weight"], G[v + "layernorm.bias"])
        del dtok
        for i in reversed(range(p.vit_layers)):
            # the bias gradient of output.dense of layer i belongs to the bucket that closes after layer i + 1 (arena layout shared
            # with the bf16 path, where it comes out of a LayerNorm backward): compute it first, then close that bucket
            # [r3] (every Linear's weight and bias gradient in one call: the weight-gradient GEMM sums the columns of the dY rows it stages)
            L = S["layers"][i]
            l = f"{v}encoder.layer.{i}."
            a = l + "attention.attention."
            T.gemm_tn_bias(h, dx, L["hid"], G[l + "output.dense.weight"], G[l + "output.dense.bias"], M, D, 4 * D)
            red.bucket_ready(ga["bounds"][p.vit_layers - 1 - i], ga["bounds"][p.vit_layers - i])
            # MLP
            dz = T.gemm_nt(h, dx, T.transpose(P[l + "output.dense.weight"]), None, M, 4 * D, D, epi=T.TE_GELU_GRAD, r=L["z"])
            T.gemm_tn_bias(h, dz, L["y2"], G[l + "intermediate.dense.weight"], G[l + "intermediate.dense.bias"], M, 4 * D, D)
            dy2 = T.gemm_nt(h, dz, T.transpose(P[l + "intermediate.dense.weight"]), None, M, D, 4 * D, epi=T.TE_NONE)
            del dz
            dxm = T.layernorm_bwd(L["xm"], dy2, P[l + "layernorm_after.weight"], L["m2"], L["r2"], G[l + "layernorm_after.weight"],
                                  G[l + "layernorm_after.bias"], dres=dx)
            # attention
            T.gemm_tn_bias(h, dxm, L["ctx"], G[l + "attention.output.dense.weight"], G[l + "attention.output.dense.bias"], M, D, D)
            dctx = T.gemm_nt(h, dxm, T.transpose(P[l + "attention.output.dense.weight"]), None, M, D, D, epi=T.TE_NONE)
            dqkv = T.attention_bwd(L["qkv"], L["ctx"], dctx, L["lse"], B, seq, heads, prec)
            del dctx
            wt = torch.empty((D, 3 * D), dtype=torch.float32, device=dev)      # [Wq^T | Wk^T | Wv^T]
            for s, nme in enumerate(("query", "key", "value")):
                T.gemm_tn_bias(h, dqkv[:, s * D:], L["y1"], G[a + nme + ".weight"], G[a + nme + ".bias"], M, D, D, ldy=3 * D)
                T.transpose(P[a + nme + ".weight"], out=wt[:, s * D:], ldo=3 * D)
            dy1 = T.gemm_nt(h, dqkv, wt, None, M, D, 3 * D, epi=T.TE_NONE)
            del dqkv
            dx = T.layernorm_bwd(L["x"], dy1, P[l + "layernorm_before.weight"], L["m1"], L["r1"], G[l + "layernorm_before.weight"],
                                 G[l + "layernorm_before.bias"], dres=dxm)
        # patch embedding: weight, position embeddings, bias / mask token
        T.gemm_tn(h, dx, S["hm"], G[v + "embeddings.patch_embeddings.projection.weight"], M, D, 256, loader=T.LD_PATCH)
        dpos = G[v + "embeddings.position_embeddings"]
        T.colsum(dx, dpos, B, seq * D)                                    # sum over the batch: dx viewed as [B, seq*D]
        _lib.check(lib.egotap_train_patch_split(h, T._p(dpos), T._p(G[v + "embeddings.patch_embeddings.projection.bias"]),
                                                T._p(G[v + "embeddings.mask_token"]), 0, st()))
        nb = len(ga["bounds"])
        red.bucket_ready(ga["bounds"][nb - 3], ga["bounds"][nb - 2])
        red.bucket_ready(ga["bounds"][nb - 2], ga["bounds"][nb - 1])
        red.finish()
        _publish_grads(P, G, held)
        ctx.egotap = None
        return (None, None) + (None,) * len(keys)


class LiftTrainBf16Fn(torch.autograd.Function):
    """The training step of the head in the bf16-STORAGE mode (EGOTAP_PREC_BF16; BASELINE configs 3-5, the wrapper's --use_amp):
    the ViT's activations live in HBM as bf16 (written by their producers: LayerNorm, the GEMM epilogues, attention), the fp32 master
    weights are rounded once per step (bf16 copy + transposed bf16 copy for the input-gradient GEMMs), every large product reads bf16
    operands through the LDS DMA (csrc/gemm_bf16s.h, gemm_tn_bf16s.h) with fp32 accumulation; the residual stream, LayerNorm / BatchNorm
    statistics, the small FC layers, the propagation units, the pose head, all gradients and the optimizer state stay fp32.
    Same autograd contract as LiftTrainFn (model/egotap_autoencoder_model.py:299-323 drives it unchanged)."""

    @staticmethod
    def _prep(net, P, dev):
        """per-step bf16 copies of the GEMM weights in a persistent arena on the module"""
        from . import bf16s as S
        p = net.preset
        D = p.vit_dim
        ar = getattr(net, "_bf16_arena", None)
        if ar is None or ar["dev"] != dev:
            bf = lambda *shape: torch.empty(shape, dtype=torch.bfloat16, device=dev)      # noqa: E731
            K1p, K1r = p.ppd * p.ppd * D, 2 * p.hm_size * p.hm_size
            ar = dict(dev=dev, layers=[dict(qkv=bf(3 * D, D), qkv_t=bf(D, 3 * D), o=bf(D, D), o_t=bf(D, D), up=bf(4 * D, D), up_t=bf(D, 4 * D),
                                            dn=bf(D, 4 * D), dn_t=bf(4 * D, D)) for _ in range(p.vit_layers)],
                      fc1p=bf(2048, K1p), fc1p_t=bf(K1p, 2048), fc1r=bf(2048, K1r))
            net._bf16_arena = ar
        v = "pos_heatmap_encoder.vit."
        for i, L in enumerate(ar["layers"]):
            l = f"{v}encoder.layer.{i}."
            for sidx, nme in enumerate(("query", "key", "value")):
                S.prep_weight(P[l + "attention.attention." + nme + ".weight"], L["qkv"][sidx * D:(sidx + 1) * D], L["qkv_t"][:, sidx * D:(sidx + 1) * D])
            S.prep_weight(P[l + "attention.output.dense.weight"], L["o"], L["o_t"])
            S.prep_weight(P[l + "intermediate.dense.weight"], L["up"], L["up_t"])
            S.prep_weight(P[l + "output.dense.weight"], L["dn"], L["dn_t"])
        S.prep_weight(P["pos_heatmap_encoder.fc1.fc.weight"], ar["fc1p"], ar["fc1p_t"])
        S.prep_weight(P["rot_heatmap_encoder.fc1.fc.weight"], ar["fc1r"])
        return ar

    @staticmethod
    def forward(ctx, net, hm, *params):
        from . import bf16s as S
        p = net.preset
        keys = _param_order(p)
        P = dict(zip(keys, params))
        dev = hm.device
        h = net._ensure_handle()
        net._bind(dev)
        B, D, seq, heads, J, T_, hid = hm.shape[0], p.vit_dim, p.seq, p.vit_heads, p.n_joints_hm, p.tokens, p.hidden
        M, BT = B * seq, B * T_
        lib = _lib.load()
        st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)      # noqa: E731
        v = "pos_heatmap_encoder.vit."
        hm = hm.detach().float().contiguous()
        W = LiftTrainBf16Fn._prep(net, P, dev)
        hm_b = S.from_f32(hm)                                       # bf16 copy of the input heatmaps: the rotation encoder's fc1 operand
        saved = {"hm": hm, "hm_b": hm_b}
        x = torch.empty((M, D), dtype=torch.float32, device=dev)
        # patch embedding on the bf16-storage GEMM: bf16 heatmaps, a per-step bf16 copy of the projection weight (as the one-call ABI does)
        pw = P[v + "embeddings.patch_embeddings.projection.weight"]
        pwb = torch.empty((D, 256), dtype=torch.bfloat16, device=dev)
        S.prep_weight(pw.detach().reshape(D, 256), pwb)
        _lib.check(lib.egotap_bf16_patch_fwd(h, T._p(hm_b), T._p(pwb), T._p(P[v + "embeddings.patch_embeddings.projection.bias"]),
                                             T._p(P[v + "embeddings.mask_token"]), T._p(P[v + "embeddings.position_embeddings"]),
                                             T._p(S.zero_page(dev)), T._p(x), B, st()))
        layers = []
        for i in range(p.vit_layers):
            l = f"{v}encoder.layer.{i}."
            a = l + "attention.attention."
            Wl = W["layers"][i]
            y1, m1, r1 = S.layernorm_fwd(x, P[l + "layernorm_before.weight"], P[l + "layernorm_before.bias"])
            bqkv = torch.cat((P[a + "query.bias"], P[a + "key.bias"], P[a + "value.bias"]))
            qkv = S.gemm_nt(y1, Wl["qkv"], bqkv)
            ctx_, lse = S.attention_fwd(qkv, B, seq, heads)
            xm = S.gemm_nt(ctx_, Wl["o"], P[l + "attention.output.dense.bias"], epi="residual", aux=x)
            y2, m2, r2 = S.layernorm_fwd(xm, P[l + "layernorm_after.weight"], P[l + "layernorm_after.bias"])
            z, hid_ = S.gemm_nt(y2, Wl["up"], P[l + "intermediate.dense.bias"], epi="gelu_save")
            xo = S.gemm_nt(hid_, Wl["dn"], P[l + "output.dense.bias"], epi="residual", aux=xm)
            layers.append(dict(x=x, m1=m1, r1=r1, y1=y1, qkv=qkv, ctx=ctx_, lse=lse, xm=xm, m2=m2, r2=r2, y2=y2, z=z, hid=hid_))
            x = xo
        tokens, mf, rf = S.layernorm_fwd(x, P[v + "layernorm.weight"], P[v + "layernorm.bias"])
        saved.update(layers=layers, xf=x, mf=mf, rf=rf, tokens=tokens)
        bufs = dict(net.named_buffers())

        def encoder(name, which, src, wb):
            acts = []
            a_in = None
            for j, n_out in enumerate((2048, 512, hid), start=1):
                f = f"{name}.fc{j}"
                if j == 1:
                    zz = S.fc1_fwd(h, which, src, wb, P[f + ".fc.bias"], B, T_)
                    K = wb.shape[1]
                else:
                    K = a_in.shape[1]
                    zz = T.gemm_nt(h, a_in, P[f + ".fc.weight"], P[f + ".fc.bias"], BT, n_out, K, epi=T.TE_BIAS)
                yy, mean, rstd = T.bn_lrelu_fwd(zz, P[f + ".bn.weight"], P[f + ".bn.bias"], bufs[f + ".bn.running_mean"], bufs[f + ".bn.running_var"])
                bufs[f + ".bn.num_batches_tracked"].add_(1)
                acts.append(dict(a_in=a_in, z=zz, y=yy, mean=mean, rstd=rstd, K=K, N=n_out, name=f))
                a_in = yy
            return acts

        pos_acts = encoder("pos_heatmap_encoder", 0, tokens, W["fc1p"])
        rot_acts = encoder("rot_heatmap_encoder", 1, hm_b, W["fc1r"])
        posz, rotz = pos_acts[-1]["y"], rot_acts[-1]["y"]
        nb, off = C.c_size_t(), C.c_size_t()
        _lib.check(lib.egotap_train_pu_saved_bytes(h, B, C.byref(nb), C.byref(off)))
        pu_saved = torch.empty(nb.value, dtype=torch.uint8, device=dev)
        _lib.check(lib.egotap_train_pu_fwd(h, T._p(posz), T._p(rotz), B, T._p(pu_saved), pu_saved.numel(), st()))
        hs1 = pu_saved[off.value: off.value + 4 * J * B * p.pu_hidden].view(torch.float32)
        pose = torch.empty((B, p.out_joints, 3), dtype=torch.float32, device=dev)
        _lib.check(lib.egotap_train_pose_head_fwd(h, T._p(posz), T._p(hs1), B, T._p(pose), st()))
        saved.update(pos_acts=pos_acts, rot_acts=rot_acts, pu_saved=pu_saved, hs1=hs1, P=P, keys=keys, net=net, B=B, W=W)
        ctx.egotap = saved
        return pose

    @staticmethod
    def backward(ctx, dpose):
        from . import bf16s as S
        Sv = ctx.egotap
        net, P, keys, B, W = Sv["net"], Sv["P"], Sv["keys"], Sv["B"], Sv["W"]
        p = net.preset
        h = net._ensure_handle()
        dev = dpose.device
        D, seq, heads, J, T_, hid, H = p.vit_dim, p.seq, p.vit_heads, p.n_joints_hm, p.tokens, p.hidden, p.pu_hidden
        M, BT = B * seq, B * T_
        lib = _lib.load()
        st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)      # noqa: E731
        ga, G = _grad_arena(net, P)                            # every entry is fully overwritten below
        held = _held_grads(P, G)
        red = net._reducer()
        red.begin(ga["flat"])
        dpose = dpose.detach().float().contiguous()
        v = "pos_heatmap_encoder.vit."
        posz, rotz, hs1 = Sv["pos_acts"][-1]["y"], Sv["rot_acts"][-1]["y"], Sv["hs1"]
        # net._stage_trace = {} (tests): every intermediate of this backward is kept under a name, together with the forward's saved
        # activations, so that each stage can be checked against its own inputs (tests/test_gpu_bf16_stages.py)
        tr = getattr(net, "_stage_trace", None)
        if tr is not None:
            tr.update(saved=Sv, W=W, dpose=dpose)
        # pose head + propagation units (fp32, as in every mode)
        dposz, drotz = torch.empty_like(posz), torch.empty_like(rotz)
        dhs1 = torch.empty(J * B * H, dtype=torch.float32, device=dev)
        gw, gb = G.get("global_mlp.pose_fcs.0.weight"), G.get("global_mlp.pose_fcs.0.bias")
        _lib.check(lib.egotap_train_pose_head_bwd(h, T._p(posz), T._p(hs1), T._p(dpose), B, T._p(dposz), T._p(dhs1),
                                                  T._p(G["pose_mlp.pose_fcs.0.weight"]), T._p(G["pose_mlp.pose_fcs.0.bias"]), T._p(gw), T._p(gb), 0, st()))
        wsb = C.c_size_t()
        _lib.check(lib.egotap_train_pu_bwd_ws_bytes(h, B, C.byref(wsb)))
        ws = torch.empty(wsb.value, dtype=torch.uint8, device=dev)
        ptrs = (C.c_void_p * 14)()
        c = "skel_sequential_layer.lstm_custom.layers."
        for i, nme in enumerate(("0.x2f", "0.x2h", "0.b2h", "0.h2h", "1.x2f", "1.x2h", "1.h2h")):
            ptrs[2 * i] = G[c + nme + ".weight"].data_ptr()
            ptrs[2 * i + 1] = G[c + nme + ".bias"].data_ptr()
        _lib.check(lib.egotap_train_pu_bwd(h, T._p(posz), T._p(rotz), B, T._p(Sv["pu_saved"]), T._p(dhs1), T._p(dposz), T._p(drotz), ptrs, 0,
                                           T._p(ws), ws.numel(), st()))
        del ws

        def encoder_bwd(acts, dy, which, src, wt):
            """fc3, fc2 in fp32 (small); fc1 on the bf16 kernels.  Returns the token gradient (position encoder) or None"""
            for j in (2, 1, 0):
                a = acts[j]
                f = a["name"]
                dz = T.bn_lrelu_bwd(a["z"], a["y"], dy, P[f + ".bn.weight"], a["mean"], a["rstd"], G[f + ".bn.weight"], G[f + ".bn.bias"])
                T.colsum(dz, G[f + ".fc.bias"], BT, a["N"])
                if j > 0:
                    T.gemm_tn(h, dz, a["a_in"], G[f + ".fc.weight"], BT, a["N"], a["K"])
                    dy = T.gemm_nt(h, dz, T.transpose(P[f + ".fc.weight"]), None, BT, a["K"], a["N"], epi=T.TE_NONE)
                else:
                    dzb = S.from_f32(dz)
                    S.fc1_wgrad(h, which, dzb, src, G[f + ".fc.weight"], B)
                    return S.fc1_dgrad_tokens(h, dzb, wt, B, seq, D) if wt is not None else None

        encoder_bwd(Sv["rot_acts"], drotz, 1, Sv["hm_b"], None)
        dtok = encoder_bwd(Sv["pos_acts"], dposz, 0, Sv["tokens"], W["fc1p_t"])       # bf16 [M, D], token order
        if tr is not None:
            tr.update(dposz=dposz, drotz=drotz, dtok=dtok)
        nl = p.vit_layers
        last = f"{v}encoder.layer.{nl - 1}."
        dx, dxb = S.layernorm_bwd(Sv["xf"], dtok, P[v + "layernorm.weight"], Sv["mf"], Sv["rf"], G[v + "layernorm.weight"], G[v + "layernorm.bias"],
                                  dcolsum=G[last + "output.dense.bias"])
        if tr is not None:
            tr["dx_final"], tr["dxb_final"] = dx, dxb
        del dtok
        for i in reversed(range(nl)):
            red.bucket_ready(ga["bounds"][nl - 1 - i], ga["bounds"][nl - i])      # everything above layer i is final: start its all-reduce
            L = Sv["layers"][i]
            Wl = W["layers"][i]
            l = f"{v}encoder.layer.{i}."
            a = l + "attention.attention."
            # MLP: dx is the output gradient of output.dense (its bias gradient came with the LayerNorm backward that produced dx)
            S.gemm_tn(dxb, L["hid"], G[l + "output.dense.weight"])
            dz = S.gemm_nt(dxb, Wl["dn_t"], None, epi="gelu_grad", aux=L["z"], colsum_out=G[l + "intermediate.dense.bias"])
            if tr is not None:
                tr[f"L{i}"] = t_ = dict(dx_in=dx, dxb_in=dxb, dz=dz)
            del dxb
            S.gemm_tn(dz, L["y2"], G[l + "intermediate.dense.weight"])
            dy2 = S.gemm_nt(dz, Wl["up_t"], None)
            del dz
            dxm, dxmb = S.layernorm_bwd(L["xm"], dy2, P[l + "layernorm_after.weight"], L["m2"], L["r2"], G[l + "layernorm_after.weight"],
                                        G[l + "layernorm_after.bias"], dres=dx, dcolsum=G[l + "attention.output.dense.bias"])
            if tr is not None:
                t_.update(dy2=dy2, dxm=dxm, dxmb=dxmb)
            del dy2, dx
            # attention
            S.gemm_tn(dxmb, L["ctx"], G[l + "attention.output.dense.weight"])
            dctx = S.gemm_nt(dxmb, Wl["o_t"], None)
            del dxmb
            dqkv = S.attention_bwd(L["qkv"], L["ctx"], dctx, L["lse"], B, seq, heads,
                                   bias_grads=tuple(G[a + nme + ".bias"] for nme in ("query", "key", "value")))
            if tr is not None:
                t_.update(dctx=dctx, dqkv=dqkv)
            del dctx
            for sidx, nme in enumerate(("query", "key", "value")):
                S.gemm_tn(dqkv[:, sidx * D:(sidx + 1) * D], L["y1"], G[a + nme + ".weight"])
            dy1 = S.gemm_nt(dqkv, Wl["qkv_t"], None)
            if tr is not None:
                t_["dy1"] = dy1
            del dqkv
            prev_bias = G[f"{v}encoder.layer.{i - 1}.output.dense.bias"] if i > 0 else None
            dx, dxb = S.layernorm_bwd(L["x"], dy1, P[l + "layernorm_before.weight"], L["m1"], L["r1"], G[l + "layernorm_before.weight"],
                                      G[l + "layernorm_before.bias"], dres=dxm, dcolsum=prev_bias, want_bf16=i > 0)
            if tr is not None:
                t_.update(dx_out=dx, dxb_out=dxb)
            del dy1, dxm
        # patch embedding (fp32 operands: the input heatmaps): weight, position embeddings, bias / mask token
        T.gemm_tn(h, dx, Sv["hm"], G[v + "embeddings.patch_embeddings.projection.weight"], M, D, 256, loader=T.LD_PATCH)
        dpos = G[v + "embeddings.position_embeddings"]
        T.colsum(dx, dpos, B, seq * D)
        _lib.check(lib.egotap_train_patch_split(h, T._p(dpos), T._p(G[v + "embeddings.patch_embeddings.projection.bias"]),
                                                T._p(G[v + "embeddings.mask_token"]), 0, st()))
        nb = len(ga["bounds"])
        red.bucket_ready(ga["bounds"][nb - 3], ga["bounds"][nb - 2])
        red.bucket_ready(ga["bounds"][nb - 2], ga["bounds"][nb - 1])
        red.finish()
        _publish_grads(P, G, held)
        ctx.egotap = None
        return (None, None) + (None,) * len(keys)


class PoseLossFn(torch.autograd.Function):
    """returns a tensor [2] = (loss_pose, loss_cos_sim) exactly as backward_AutoEncoder weighs them"""

    @staticmethod
    def forward(ctx, net, pred, gt, lambda_mpjpe, lambda_cos_sim):
        out, dpred = T.pose_loss(net._ensure_handle(), pred.detach().contiguous(), gt.detach().float().contiguous(), lambda_mpjpe, lambda_cos_sim)
        ctx.save_for_backward(dpred)
        return out

    @staticmethod
    def backward(ctx, dout):
        (dpred,) = ctx.saved_tensors
        # the kernel keeps the two terms apart, so any weighting of (loss_pose, loss_cos_sim) downstream gets its exact gradient
        return None, dpred[0] * dout[0] + dpred[1] * dout[1], None, None, None


def lift_train_forward(net, hm):
    """training-mode forward of EgoTAPAutoEncoder through the HIP operators, differentiable w.r.t. net.parameters()"""
    if net.preset.seq % 32 != 0:
        raise NotImplementedError(f"training the lifting head needs a ViT sequence that is a multiple of 32 (heatmap sides 64, 128: every shipped script); "
                                  f"--load_size_heatmap {net.preset.hm_size} gives {net.preset.seq} tokens -- evaluation runs at any side that is a multiple of 16")
    params = dict(net.named_parameters())
    # net.bf16_storage = False keeps fp32 tensors in HBM under the bf16 arithmetic (round 1's path: operands converted per launch)
    bf16 = getattr(net, "precision", "f32") == "bf16"
    bf16s = bf16 and net.preset.vit_dim == 1024 and getattr(net, "bf16_storage", True)
    if getattr(net, "one_call_training", True) and (bf16s or not bf16 or net.preset.vit_dim != 1024):
        fn = LiftTrainOneCallFn        # the library picks the bf16-storage step itself (precision bf16, vit_dim 1024)
    else:
        fn = LiftTrainBf16Fn if bf16s else LiftTrainFn
    return fn.apply(net, hm, *[params[k] for k in _param_order(net.preset)])


class EgotapAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction) with the update done by the HIP kernels.
    When the gradients of a group are views of ONE flat arena (the lifting head's training Functions publish them that way) the
    whole group is updated by a single launch (egotap_train_adamw_multi): the moment buffers become views of two flat arenas with the
    gradient arena's layout; otherwise one launch per tensor.  state_dict() keeps torch.optim.AdamW's format either way."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = {}

    def _flat_plan(self, gi, group):
        """(table, g_base, m_flat, v_flat, span, segs) if every gradient of the group lies in one allocation, else None"""
        ps = [p for p in group["params"] if p.grad is not None]
        if len(ps) < 2 or not all(p.grad.is_cuda and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and p.is_contiguous() for p in ps):
            return None
        base = min(p.grad.untyped_storage().data_ptr() for p in ps)
        if any(p.grad.untyped_storage().data_ptr() != base for p in ps):
            return None
        sig = tuple((p.data_ptr(), p.grad.data_ptr(), p.numel()) for p in ps)
        plan = self._flat.get(gi)
        if plan is not None and plan["sig"] == sig:
            return plan
        segs = sorted(((p.grad.data_ptr() - base) // 4, p.numel(), p) for p in ps)
        span = segs[-1][0] + segs[-1][1]
        dev = ps[0].device
        old = plan
        m_flat = torch.zeros(span, dtype=torch.float32, device=dev)
        v_flat = torch.zeros(span, dtype=torch.float32, device=dev)
        table = torch.tensor([[o, n, p.data_ptr()] for o, n, p in segs], dtype=torch.int64, device=dev)
        g_flat = torch.empty(0, dtype=torch.float32, device=dev).set_(ps[0].grad.untyped_storage(), 0, (span,))
        plan = dict(sig=sig, segs=segs, span=span, m=m_flat, v=v_flat, table=table, g=g_flat)
        self._flat[gi] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            plan = self._flat_plan(gi, group)
            if plan is not None:
                steps = set()
                for o, n, p in plan["segs"]:
                    st = self.state[p]
                    mv, vv = plan["m"][o:o + n].view(p.shape), plan["v"][o:o + n].view(p.shape)
                    if not st:
                        st["step"] = 0
                    else:                                   # state from a checkpoint or from per-tensor steps: move it into the arenas
                        if st["exp_avg"].data_ptr() != mv.data_ptr():
                            mv.copy_(st["exp_avg"])
                            vv.copy_(st["exp_avg_sq"])
                    st["exp_avg"], st["exp_avg_sq"] = mv, vv
                    st["step"] = int(st["step"]) + 1
                    steps.add(st["step"])
                if len(steps) == 1:
                    _lib.check(_lib.load().egotap_train_adamw_multi(T._p(plan["table"]), len(plan["segs"]), T._p(plan["g"]), T._p(plan["m"]), T._p(plan["v"]),
                                                                    plan["span"], group["lr"], b1, b2, group["eps"], group["weight_decay"], steps.pop(), T._s()))
                    continue
                for o, n, p in plan["segs"]:                # tensors at different step counts (partial checkpoints): per tensor
                    st = self.state[p]
                    T.adamw(p, p.grad, st["exp_avg"], st["exp_avg_sq"], group["lr"], st["step"], b1, b2, group["eps"], group["weight_decay"])
                continue
            for p in group["params"]:
                if p.grad is None:
                    continue                        # cls_token / pooler: never receive a gradient, never move
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] = int(st["step"]) + 1        # torch.optim.AdamW checkpoints keep the step as a tensor
                T.adamw(p, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"], group["lr"], st["step"], b1, b2, group["eps"],
                        group["weight_decay"])
        return None


def get_scheduler(optimizer, opt):
    """Learning-rate policies of the reference (model/network.py:35-55), stepped once per iteration by
    ``update_learning_rate`` (train.py:130).  ``cos_anneal_warmup`` restates transformers.optimization
    .get_cosine_schedule_with_warmup (linear warm-up over niter epochs, then half a cosine to zero over niter_decay epochs):
    the reference imports it; here it is a LambdaLR with the same multiplier, pinned by tests/golden/lr_schedules.npz."""
    import math
    from torch.optim import lr_scheduler
    policy = getattr(opt, "lr_policy", "lambda")
    niter, niter_decay = getattr(opt, "niter", 0), getattr(opt, "niter_decay", 0)
    per_epoch = getattr(opt, "epoch_iter_cnt", 1)
    if policy == "lambda":
        epoch_count = getattr(opt, "epoch_count", 1)
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda e: 1.0 - max(0, e + epoch_count - niter) / float(niter_decay + 1))
    if policy == "step":
        return lr_scheduler.StepLR(optimizer, step_size=getattr(opt, "lr_decay_iters_step", 4), gamma=0.5)
    if policy == "exponent":
        return lr_scheduler.ExponentialLR(optimizer, gamma=0.95)
    if policy == "cos_anneal":
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=(niter + niter_decay) * per_epoch)
    if policy == "cos_anneal_warmup":
        warm, total = niter * per_epoch, (niter + niter_decay) * per_epoch

        def mult(step):
            if step < warm:
                return float(step) / float(max(1, warm))
            progress = float(step - warm) / float(max(1, total - warm))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 2.0 * 0.5 * progress)))
        return lr_scheduler.LambdaLR(optimizer, mult)
    raise NotImplementedError("learning rate policy [%s] is not implemented" % policy)
