"""Host-side mirror of the reference's network modules for the hot path.

Same class names, constructor arguments, forward signatures and ``state_dict`` keys as
``model/net_architecture.py`` in the reference, so ``train.py``/``test.py``-style callers and released
checkpoints work unchanged -- but the modules hold parameters only: every forward goes through the
C ABI of libegotap_hip.so (hand-written HIP for gfx950).  There is no PyTorch compute fallback; without
the HIP library or off the GPU these modules raise.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.nn as nn

from . import lib as _lib
from . import spec as _spec


class _Node(nn.Module):
    """Parameter container; the tree of _Nodes reproduces the reference's dotted state_dict keys."""

    def forward(self, *a, **k):  # pragma: no cover - containers are never called
        raise RuntimeError("parameter container: compute happens in libegotap_hip.so")


def _build_tree(root: nn.Module, entries):
    """Register (key, shape) entries as nn.Parameter / buffers under nested _Node children of root."""
    for key, shape in entries:
        parts = key.split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, _Node())
            mod = mod._modules[p]
        leaf = parts[-1]
        if leaf == "num_batches_tracked":
            mod.register_buffer(leaf, torch.zeros(shape, dtype=torch.long))
        elif _spec.is_buffer(key):
            mod.register_buffer(leaf, torch.ones(shape) if leaf == "running_var" else torch.zeros(shape))
        else:
            mod.register_parameter(leaf, nn.Parameter(torch.zeros(shape)))


def _kaiming_init_(module: nn.Module):
    """init_net(net, 'kaiming') of the reference (network_utils.py:37-58): kaiming-normal (fan_in) on every
    Conv/Linear weight, zero bias; everything else keeps its constructor default.  Uses torch's RNG."""
    for name, p in module.named_parameters():
        leaf = name.rsplit(".", 1)[-1]
        with torch.no_grad():
            if leaf == "weight" and p.dim() >= 2:
                nn.init.kaiming_normal_(p, a=0, mode="fan_in")
            elif leaf == "weight":            # LayerNorm / BatchNorm1d gain
                p.fill_(1.0)
            elif leaf == "bias":
                p.zero_()
            elif leaf in ("cls_token", "position_embeddings"):
                nn.init.trunc_normal_(p, mean=0.0, std=0.02)
            elif leaf == "mask_token":
                p.zero_()


class EgoTAPAutoEncoder(nn.Module):
    """Heatmaps -> 3D pose lifting head (reference: model/net_architecture.py:579-758).

    forward(input[B, 6J, S, S]) -> (pose[B, J(+1), 3], rot zeros[B, 3J], indep_pos zeros[B, 6J],
    reconstructed-heatmap zeros[B, 6J, S, S]) -- the last three are all-zero in the reference too
    (net_architecture.py:718-719, 756); they are returned as cached / broadcast zeros, not re-allocated.
    """

    def __init__(self, opt, input_channel_scale: int = 2, fc_dim: int = 16384):
        super().__init__()
        if input_channel_scale != 2:
            raise NotImplementedError("only the stereo presets (UnrealEgo, EgoCap) are built")
        if not getattr(opt, "patched_heatmap_ae", True) or getattr(opt, "skel_layer", "PU") != "PU":
            raise NotImplementedError("only --patched_heatmap_ae --skel_layer PU (the shipped configuration) is built")
        if getattr(opt, "heatmap_type", "sin") != "sin":
            raise NotImplementedError("only --heatmap_type sin is built")
        hm = list(getattr(opt, "load_size_heatmap", [64, 64]))
        if hm[0] != hm[1]:
            raise ValueError("load_size_heatmap must be square")
        self.preset = _spec.lift_preset(opt.joint_preset, hm[0], getattr(opt, "ae_hidden_size", 128))
        if opt.num_heatmap != self.preset.n_joints_hm or opt.num_rot_heatmap != self.preset.n_joints_hm:
            raise ValueError("num_heatmap / num_rot_heatmap must match the joint preset")
        p = self.preset
        self.joint_preset = opt.joint_preset
        self.hidden_size = p.hidden
        self.num_joints = p.out_joints
        self.num_pos_heatmap = self.num_rot_heatmap = p.n_joints_hm
        self.channels_heatmap = p.in_channels
        self.W = self.H = p.hm_size
        self.rot_dim = 3 * p.n_joints_hm
        _build_tree(self, _spec.lift_state_spec(p))
        _kaiming_init_(self)
        self._handle = None
        self._bound_sig = None
        self._ws = None
        self._zeros = {}
        # several processes on one GPU (rehearsals, tests): per-step propagation-unit kernels instead of the one-launch recurrence.  An OPTION of the
        # model (opt.shared_device); the EGOTAP_SHARED_DEVICE=1 environment switch remains for launchers that cannot reach opt
        self._shared_device = bool(getattr(opt, "shared_device", False))

    # -- C-ABI plumbing ---------------------------------------------------------------------------
    def _ensure_handle(self):
        if self._handle is None:
            p = self.preset
            cfg = _lib.EgotapConfig(C.sizeof(_lib.EgotapConfig), p.n_joints_hm, int(p.estimate_head), p.hm_size, p.hidden,
                                    p.vit_dim, p.vit_heads, p.vit_layers, p.patch, p.pu_hidden)
            h = C.c_void_p()
            _lib.check(_lib.load().egotap_create(C.byref(cfg), C.byref(h)))
            self._handle = h
            if self._shared_device or os.environ.get("EGOTAP_SHARED_DEVICE", "0") == "1":      # several processes on one GPU: see set_pu_chain
                _lib.check(_lib.load().egotap_set_pu_chain(h, 0))
        return self._handle

    def set_pu_chain(self, enable: bool = True):
        """The propagation units' recurrence as one launch per layer (default; its workgroups wait for each other, so the process must
        have the GPU to itself while a forward runs) or as one kernel per step (enable = False: safe on a shared device, same bits).
        EGOTAP_SHARED_DEVICE=1 in the environment selects the per-step kernels for every module of the process."""
        _lib.check(_lib.load().egotap_set_pu_chain(self._ensure_handle(), int(bool(enable))))
        return self

    def pu_chain_status(self):
        """(enabled, faults): whether this module still runs the recurrence as one launch per layer, and how many of its launches had
        to be redone on the device because their workgroups were not co-resident (egotap.h egotap_pu_chain_status; results were
        right every time -- the library switches itself to the per-step kernels after the first).  Exact after a synchronise."""
        en, nf = C.c_int(), C.c_int()
        _lib.check(_lib.load().egotap_pu_chain_status(self._ensure_handle(), C.byref(en), C.byref(nf)))
        return bool(en.value), nf.value

    def _bind(self, device):
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        sig = tuple((k, t.data_ptr()) for k, t in sd.items())
        if sig == self._bound_sig:
            return
        lib, h = _lib.load(), self._ensure_handle()
        for k, t in sd.items():
            if t.device != device:
                raise _lib.EgotapError(f"parameter {k} is on {t.device}, input on {device}")
            if t.dtype == torch.float32:
                if not t.is_contiguous():
                    raise _lib.EgotapError(f"parameter {k} must be contiguous")
                _lib.check(lib.egotap_bind_param(h, _lib.NET_LIFT, k.encode(), C.c_void_p(t.data_ptr()), t.numel(), _lib.F32))
            elif t.dtype == torch.long:
                _lib.check(lib.egotap_bind_param(h, _lib.NET_LIFT, k.encode(), C.c_void_p(t.data_ptr()), t.numel(), _lib.I64))
            else:
                raise _lib.EgotapError(f"parameter {k}: dtype {t.dtype} not supported (fp32 path)")
        n = C.c_int()
        _lib.check(lib.egotap_unbound_count(h, _lib.NET_LIFT, C.byref(n)))
        if n.value:
            raise _lib.EgotapError(f"{n.value} parameters the forward needs are not bound")
        self._bound_sig = sig

    def _workspace(self, B, device):
        lib, h = _lib.load(), self._ensure_handle()
        need = C.c_size_t()
        _lib.check(lib.egotap_lift_workspace_bytes(h, B, C.byref(need)))
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != device:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=device)
        self._act_scratch(B, device)
        return self._ws

    def _act_scratch(self, B, device):
        """bf16 mode: scratch for the bf16 copy of a GEMM's activation operand (egotap_set_act_scratch), sized for the ViT MLP's
        hidden activations [B * seq, 4 * D] and grown with the batch"""
        if getattr(self, "precision", "f32") != "bf16" or device.type != "cuda":
            return
        if B * self.preset.seq >= 4096:      # the bf16-storage forward (egotap_lift_forward at batches that fill the chip) converts nothing
            return
        need = 2 * B * self.preset.seq * 4 * self.preset.vit_dim
        cur = getattr(self, "_ascratch", None)
        if cur is None or cur.numel() < need or cur.device != device:
            self._ascratch = torch.empty(need, dtype=torch.uint8, device=device)
            _lib.check(_lib.load().egotap_set_act_scratch(self._ensure_handle(), C.c_void_p(self._ascratch.data_ptr()), self._ascratch.numel()))

    def _reducer(self):
        """the overlapped gradient reducer of this module's training Function (egotap_amd.parallel.GradReducer; a no-op for one rank)"""
        if getattr(self, "_grad_reducer", None) is None:
            from .parallel import GradReducer
            self._grad_reducer = GradReducer()
        return self._grad_reducer

    def set_precision(self, mode: str = "f32"):
        """Arithmetic of the large GEMMs (egotap.h egotap_set_precision): "f32" = exact fp32 MFMA (default),
        "bf16x3" = fp32 operands split into hi + lo bf16, three bf16 MFMAs per product, fp32 accumulate (opt-in fast mode)."""
        if mode not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}")
        _lib.check(_lib.load().egotap_set_precision(self._ensure_handle(), _lib.PRECISIONS[mode]))
        self.precision = mode
        if mode == "bf16":             # scratch for the bf16 copy of a GEMM's weights (largest: fc1 of the position encoder)
            dev = next(self.parameters()).device
            if dev.type == "cuda" and (getattr(self, "_wscratch", None) is None or self._wscratch.device != dev):
                need = 2 * max(p.numel() for p in self.parameters() if p.dim() >= 2)
                self._wscratch = torch.empty(need, dtype=torch.uint8, device=dev)
            if getattr(self, "_wscratch", None) is not None:
                _lib.check(_lib.load().egotap_set_weight_scratch(self._ensure_handle(), C.c_void_p(self._wscratch.data_ptr()), self._wscratch.numel()))
        else:                          # the activation scratch is (re)attached by the next forward in bf16 mode (_act_scratch)
            self._ascratch = None
            _lib.check(_lib.load().egotap_set_act_scratch(self._ensure_handle(), None, 0))
        return self

    def intermediate(self, name: str, B: int):
        """View of an intermediate of the LAST forward inside the workspace (parity tests)."""
        off, n = C.c_size_t(), C.c_int64()
        _lib.check(_lib.load().egotap_lift_intermediate(self._ensure_handle(), B, name.encode(), C.byref(off), C.byref(n)))
        return self._ws[off.value: off.value + 4 * n.value].view(torch.float32)

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.load().egotap_destroy(self._handle)
        except Exception:
            pass

    # -- reference API ----------------------------------------------------------------------------
    def predict_pose(self, input, input_rgb_left=None, input_rgb_right=None):
        return self.forward(input, input_rgb_left, input_rgb_right, pose_only=True)

    def forward(self, input, input_rgb_left=None, input_rgb_right=None, pose_only=False):
        p = self.preset
        if not input.is_cuda:
            raise _lib.EgotapError("EgoTAPAutoEncoder runs on the GPU only (no CPU fallback); move the input to cuda")
        if input.dim() != 4 or input.shape[1] != p.in_channels or input.shape[2] != p.hm_size or input.shape[3] != p.hm_size:
            raise ValueError(f"expected input [B, {p.in_channels}, {p.hm_size}, {p.hm_size}], got {tuple(input.shape)}")
        if self.training:
            from .training import lift_train_forward          # training mode: batch-statistics BatchNorm, differentiable
            pose = lift_train_forward(self, input)
            return pose if pose_only else (pose,) + self._zero_outputs(input.shape[0], input.device)[1:]
        hm = input.detach()
        if hm.dtype != torch.float32:
            hm = hm.float()
        hm = hm.contiguous()
        B = hm.shape[0]
        dev = hm.device
        pose = torch.empty((B, p.out_joints, 3), dtype=torch.float32, device=dev)
        if B > 0:
            with torch.cuda.device(dev):
                self._bind(dev)
                ws = self._workspace(B, dev)
                _lib.check(_lib.load().egotap_lift_forward(
                    self._ensure_handle(), C.c_void_p(hm.data_ptr()), B, C.c_void_p(pose.data_ptr()),
                    C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        if pose_only:
            return pose
        return (pose,) + self._zero_outputs(B, dev)[1:]

    def predict_pose_graphed(self, input):
        """predict_pose through a captured HIP graph (serving at small batches, where the ~130 launches of a forward cost more than
        the kernels): egotap_lift_forward is captured once per (batch, device, precision, parameter pointers) with a static input
        and output buffer and replayed afterwards -- same kernels, same bits (tests/test_gpu_lift.py).  Eval mode only; the returned
        tensor is the graph's static output buffer (valid until the next call with the same batch)."""
        if self.training:
            raise RuntimeError("predict_pose_graphed is an inference path: call .eval() first")
        p = self.preset
        if not input.is_cuda:
            raise _lib.EgotapError("EgoTAPAutoEncoder runs on the GPU only (no CPU fallback); move the input to cuda")
        if input.dim() != 4 or tuple(input.shape[1:]) != (p.in_channels, p.hm_size, p.hm_size):
            raise ValueError(f"expected input [B, {p.in_channels}, {p.hm_size}, {p.hm_size}], got {tuple(input.shape)}")
        dev = input.device
        self._bind(dev)
        B = input.shape[0]
        key = (B, str(dev), getattr(self, "precision", "f32"), self._bound_sig)
        graphs = self.__dict__.setdefault("_graphs", {})
        g = graphs.get(key)
        if g is None:
            # Every pointer a captured launch takes is baked into the graph, so the graph OWNS what it replays into: a workspace of
            # its own (the module's grow-only self._ws is replaced -- and the old one handed back to the allocator -- as soon as a
            # larger batch arrives) and references to the bf16 scratch buffers attached to the handle at capture time (replaced, not
            # resized, when they grow; their contents are transient within one launch, so sharing them between graphs is fine).
            lib, h = _lib.load(), self._ensure_handle()
            static_in = input.detach().float().contiguous().clone()
            static_out = torch.empty((B, p.out_joints, 3), dtype=torch.float32, device=dev)
            self.predict_pose(static_in)                   # eager once: lazy occupancy queries and allocations happen here
            need = C.c_size_t()
            _lib.check(lib.egotap_lift_workspace_bytes(h, B, C.byref(need)))
            ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
            self._act_scratch(B, dev)
            keep = (ws, getattr(self, "_ascratch", None), getattr(self, "_wscratch", None))

            def run():
                _lib.check(lib.egotap_lift_forward(h, C.c_void_p(static_in.data_ptr()), B, C.c_void_p(static_out.data_ptr()),
                                                   C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.device(dev), torch.cuda.stream(side):
                run()
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.device(dev), torch.cuda.graph(graph, stream=side):
                run()
            if len(graphs) >= 8:                           # a handful of serving batch sizes; drop the oldest beyond that
                graphs.pop(next(iter(graphs)))
            g = graphs[key] = (graph, static_in, static_out, keep)
        graph, static_in, static_out, _ = g
        static_in.copy_(input)
        graph.replay()
        return static_out

    def _zero_outputs(self, B, dev):
        """(None, rot, indep_pos, reconstructed heatmaps): the reference's all-zero outputs, cached / broadcast"""
        p = self.preset
        key = (B, str(dev))
        if key not in self._zeros:
            z = torch.zeros((), dtype=torch.float32, device=dev)
            self._zeros = {key: (torch.zeros((B, self.rot_dim), device=dev), torch.zeros((B, 6 * p.n_joints_hm), device=dev),
                                 z.expand(B, p.in_channels, p.hm_size, p.hm_size))}
        rot, indep, out_hm = self._zeros[key]
        return None, rot, indep, out_hm


class HeatMap_UnrealEgo_Shared(nn.Module):
    """Stereo heatmap estimator (reference: model/net_architecture.py:25-173 over torchvision resnet18).

    forward(left[B,3,256,256], right[B,3,256,256]) -> [B, 2*n_hm, 64, 64] (left maps then right maps).
    The state_dict has the reference's 258 keys, including the duplicate ``backbone.backbone.layerK.*`` views of
    the ResNet tensors (the same nn.Parameter objects registered under both paths, as in the reference).
    ``forward_into(left, right, out)`` writes the result into a channel slice of a larger tensor instead
    (used by the wrapper to build the lifting head's input without torch.cat).
    """

    def __init__(self, opt, model_name: str = "resnet18", input_channel_scale: int = 2):
        super().__init__()
        self.model_name = model_name
        # resnet18 / resnet34 (BasicBlocks: one-call C forward, bf16 modes, stage-1 training) or resnet50 / resnet101 (Bottleneck blocks,
        # feature_scale 4: fp32 eval forward composed from the operator entry points, _forward_bottleneck)
        self.bottleneck = _spec.hm_is_bottleneck(model_name)
        self.blocks = _spec.hm_all_blocks(model_name)
        if input_channel_scale != 2:
            raise NotImplementedError("only the stereo presets are built")
        limb = {"none": 0, "sin": 2, "limb": 1}[getattr(opt, "heatmap_type", "none")]
        self.num_heatmap = opt.num_heatmap + opt.num_rot_heatmap * limb
        hm = list(getattr(opt, "load_size_heatmap", [64, 64]))
        self.hm_size = hm[0]
        self.preset = _spec.lift_preset(opt.joint_preset, hm[0], getattr(opt, "ae_hidden_size", 128))
        J = self.preset.n_joints_hm
        if self.num_heatmap == J:
            self._net = _lib.NET_HM_POS
        elif self.num_heatmap == 2 * J:
            self._net = _lib.NET_HM_ROT
        else:
            raise ValueError("heatmap estimator must be the position net (num_rot_heatmap=0) or the sin/cos net (num_heatmap=0)")
        entries = _spec.hm_state_spec(self.num_heatmap, model_name)
        _build_tree(self, [(k, s) for k, s, a in entries if a is None])
        for k, s, a in entries:                      # aliases: same Parameter / buffer object under a second path
            if a is None:
                continue
            src_mod, src_leaf = self._locate(a)
            parts = k.split(".")
            mod = self
            for part in parts[:-1]:
                if part not in mod._modules:
                    mod.add_module(part, _Node())
                mod = mod._modules[part]
            if src_leaf in src_mod._parameters:
                mod.register_parameter(parts[-1], src_mod._parameters[src_leaf])
            else:
                mod.register_buffer(parts[-1], src_mod._buffers[src_leaf])
        self._aliases = [(k, a) for k, s, a in entries if a is not None]
        _kaiming_init_(self)
        self._handle = None
        self._bound_sig = None
        self._ws = None

    def _apply(self, fn, *args, **kwargs):
        """Module._apply (.to / .cuda / .float ...) replaces every registered BUFFER by fn(buffer) -- once per registration, so the
        ``backbone.backbone.layerK.*`` aliases of the BatchNorm statistics would stop being the tensors the forward updates (parameters keep
        their identity: torch swaps .data).  In the reference the aliases are the same nn.BatchNorm2d MODULES (net_architecture.py:68-73), so they
        can never drift apart, and its load_state_dict reads the alias keys LAST: a checkpoint written with stale aliases would load stale
        statistics there.  Re-tie them after every _apply."""
        super()._apply(fn, *args, **kwargs)
        for k, a in getattr(self, "_aliases", ()):
            src_mod, src_leaf = self._locate(a)
            mod, leaf = self._locate(k)
            if src_leaf in src_mod._buffers:
                mod._buffers[leaf] = src_mod._buffers[src_leaf]
        self._bound_sig = None
        return self

    def set_precision(self, mode: str = "f32"):
        """Arithmetic of the 3x3 stride-1 convolutions with >= 128 output channels (88 % of the FLOPs): "f32" exact (default),
        "bf16x3" split operands, "bf16" rounded operands; everything else stays fp32 (egotap.h egotap_set_precision)."""
        if mode not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}")
        if self.bottleneck and mode != "f32":
            raise NotImplementedError(f"backbone {self.model_name!r} runs in fp32 only (the bf16 modes cover resnet18 / resnet34)")
        _lib.check(_lib.load().egotap_set_precision(self._ensure_handle(), _lib.PRECISIONS[mode]))
        self.precision = mode
        return self

    def _locate(self, key):
        parts = key.split(".")
        mod = self
        for part in parts[:-1]:
            mod = mod._modules[part]
        return mod, parts[-1]

    def _ensure_handle(self):
        if self._handle is None:
            p = self.preset
            # Bottleneck nets never bind to the handle's one-call forward (it only carries the operator calls): default block counts
            cfg = _lib.EgotapConfig(C.sizeof(_lib.EgotapConfig), p.n_joints_hm, int(p.estimate_head), p.hm_size, p.hidden,
                                    p.vit_dim, p.vit_heads, p.vit_layers, p.patch, p.pu_hidden,
                                    (C.c_int32 * 4)(*((2, 2, 2, 2) if self.bottleneck else self.blocks)))
            h = C.c_void_p()
            _lib.check(_lib.load().egotap_create(C.byref(cfg), C.byref(h)))
            self._handle = h
        return self._handle

    def _bind(self, device):
        sd = self.state_dict(keep_vars=True)
        sig = tuple((k, t.data_ptr()) for k, t in sd.items())
        if sig == self._bound_sig:
            return
        lib, h = _lib.load(), self._ensure_handle()
        for k, t in sd.items():
            if t.device != device:
                raise _lib.EgotapError(f"parameter {k} is on {t.device}, input on {device}")
            dt = _lib.F32 if t.dtype == torch.float32 else (_lib.I64 if t.dtype == torch.long else None)
            if dt is None or not t.is_contiguous():
                raise _lib.EgotapError(f"parameter {k}: need contiguous fp32 (or int64 counters)")
            _lib.check(lib.egotap_bind_param(h, self._net, k.encode(), C.c_void_p(t.data_ptr()), t.numel(), dt))
        n = C.c_int()
        _lib.check(lib.egotap_unbound_count(h, self._net, C.byref(n)))
        if n.value:
            raise _lib.EgotapError(f"{n.value} parameters the forward needs are not bound")
        self._bound_sig = sig

    def _workspace(self, B, device):
        need = C.c_size_t()
        _lib.check(_lib.load().egotap_hm_workspace_bytes(self._ensure_handle(), B, C.byref(need)))
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != device:
            self._ws = None
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=device)
        return self._ws

    def intermediate(self, name: str, B: int):
        off, n = C.c_size_t(), C.c_int64()
        _lib.check(_lib.load().egotap_hm_intermediate(self._ensure_handle(), B, name.encode(), C.byref(off), C.byref(n)))
        return self._ws[off.value: off.value + 4 * n.value].view(torch.float32)

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.load().egotap_destroy(self._handle)
        except Exception:
            pass

    def forward_into(self, left, right, out, channel_offset: int = 0, workspace=None):
        """Write this net's [B, 2*n_hm, S, S] output into out[:, channel_offset : channel_offset + 2*n_hm]."""
        if self.training:
            raise NotImplementedError("forward_into writes the eval-mode result (folded BatchNorm) into a caller's slice; in train mode call the module itself (differentiable path) or .eval() first")
        for t in (left, right, out):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise _lib.EgotapError("heatmap estimator needs contiguous float32 CUDA tensors (no CPU fallback)")
        B, S = left.shape[0], 4 * self.hm_size
        if tuple(left.shape) != (B, 3, S, S) or tuple(right.shape) != (B, 3, S, S):
            raise ValueError(f"expected left/right [B, 3, {S}, {S}], got {tuple(left.shape)} / {tuple(right.shape)}")
        n_out = 2 * self.num_heatmap
        if out.dim() != 4 or out.shape[0] != B or out.shape[2] != self.hm_size or out.shape[3] != self.hm_size \
                or channel_offset + n_out > out.shape[1]:
            raise ValueError("output tensor does not hold the requested channel slice")
        if B == 0:
            return out
        dev = left.device
        if self.bottleneck:
            with torch.cuda.device(dev):
                self._forward_bottleneck(left, right, out, channel_offset)
            return out
        with torch.cuda.device(dev):
            self._bind(dev)
            ws = workspace if workspace is not None else self._workspace(B, dev)
            self._ws = ws
            hw = self.hm_size * self.hm_size
            _lib.check(_lib.load().egotap_hm_forward(
                self._ensure_handle(), self._net, C.c_void_p(left.data_ptr()), C.c_void_p(right.data_ptr()), B,
                C.c_void_p(out.data_ptr() + 4 * channel_offset * hw), out.shape[1] * hw, C.c_void_p(ws.data_ptr()),
                ws.numel(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return out

    @torch.no_grad()
    def forward_bnbatch_into(self, left, right, out, channel_offset: int = 0, chunk: int = 256, workspace=None):
        """forward_into with BATCH-statistics BatchNorm2d and no graph (egotap.h egotap_hm_forward_bnbatch): what a FROZEN estimator computes
        while the lifting head trains under train.py:91 model.train() -- per-eye statistics, running_mean / running_var /
        num_batches_tracked of every BatchNorm updated twice (left, right).  bf16 precision only (the bf16 channels-last kernels); the
        backbone runs over the whole batch, the decoder in pieces of `chunk` frames.  The module's own .training flag is not consulted."""
        if self.bottleneck:
            raise NotImplementedError(f"backbone {self.model_name!r}: no batch-statistics forward (resnet18 / resnet34 have one)")
        if getattr(self, "precision", "f32") != "bf16":
            raise _lib.EgotapError("forward_bnbatch_into runs on the bf16 channels-last kernels: set_precision('bf16') first "
                                   "(fp32 / bf16x3: hm_training.hm_train_forward_nograd)")
        for t in (left, right, out):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise _lib.EgotapError("heatmap estimator needs contiguous float32 CUDA tensors (no CPU fallback)")
        B, S = left.shape[0], 4 * self.hm_size
        if tuple(left.shape) != (B, 3, S, S) or tuple(right.shape) != (B, 3, S, S):
            raise ValueError(f"expected left/right [B, 3, {S}, {S}], got {tuple(left.shape)} / {tuple(right.shape)}")
        n_out = 2 * self.num_heatmap
        if out.dim() != 4 or out.shape[0] != B or out.shape[2] != self.hm_size or out.shape[3] != self.hm_size \
                or channel_offset + n_out > out.shape[1]:
            raise ValueError("output tensor does not hold the requested channel slice")
        if B < 2:
            raise ValueError("batch-statistics BatchNorm needs at least two frames")
        dev = left.device
        chunk = B if chunk is None or chunk <= 0 else min(int(chunk), B)
        with torch.cuda.device(dev):
            self._bind(dev)
            lib, h = _lib.load(), self._ensure_handle()
            need = C.c_size_t()
            _lib.check(lib.egotap_hm_forward_bnbatch_workspace_bytes(h, B, chunk, C.byref(need)))
            ws = workspace
            if ws is None or ws.numel() < need.value or ws.device != dev:
                cur = getattr(self, "_ws_bn", None)
                if cur is None or cur.numel() < need.value or cur.device != dev:
                    self._ws_bn = None
                    self._ws_bn = torch.empty(need.value, dtype=torch.uint8, device=dev)
                ws = self._ws_bn
            hw = self.hm_size * self.hm_size
            _lib.check(lib.egotap_hm_forward_bnbatch(
                h, self._net, C.c_void_p(left.data_ptr()), C.c_void_p(right.data_ptr()), B,
                C.c_void_p(out.data_ptr() + 4 * channel_offset * hw), out.shape[1] * hw, chunk, C.c_void_p(ws.data_ptr()), ws.numel(),
                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return out

    def bnbatch_intermediate(self, name: str, B: int, chunk: int, ws=None):
        """bf16 view [B * s * s, 2 C] of a backbone map of the LAST forward_bnbatch_into inside its workspace (parity tests)"""
        off, n = C.c_size_t(), C.c_int64()
        _lib.check(_lib.load().egotap_hm_forward_bnbatch_intermediate(self._ensure_handle(), B, min(chunk, B) if chunk else B, name.encode(), C.byref(off), C.byref(n)))
        ws = ws if ws is not None else self._ws_bn
        return ws[off.value: off.value + 2 * n.value].view(torch.bfloat16)

    def bnbatch_workspace(self, B, chunk, device):
        """scratch of forward_bnbatch_into for (B, chunk), kept on the module (both estimators of a wrapper can share one)"""
        need = C.c_size_t()
        _lib.check(_lib.load().egotap_hm_forward_bnbatch_workspace_bytes(self._ensure_handle(), B, min(chunk, B) if chunk else B, C.byref(need)))
        cur = getattr(self, "_ws_bn", None)
        if cur is None or cur.numel() < need.value or cur.device != device:
            self._ws_bn = None
            self._ws_bn = torch.empty(need.value, dtype=torch.uint8, device=device)
        return self._ws_bn

    @torch.no_grad()
    def _forward_bottleneck(self, left, right, out, channel_offset):
        """Eval forward of the resnet50 / resnet101 estimators (net_architecture.py:45-51 backbone once per eye, :75-85 pyramid, :139-173
        decoder on the channel-concatenated eyes), composed from the library's operator entry points: every convolution is one
        conv_f32 kernel launch with its BatchNorm (eval) / bias, residual and ReLU in the epilogue; images n = 2b + eye, so a stage output
        [2B, C, s, s] IS the stereo concat [B, 2C, s, s]; upsamples and 1x1 skips write channel slices of the concat buffers in place."""
        from . import hm_ops as H
        h = self._ensure_handle()
        sd = {k: v for k, v in self.state_dict(keep_vars=True).items()}
        for k, t in sd.items():
            if t.device != left.device or not t.is_contiguous() or t.dtype not in (torch.float32, torch.long):
                raise _lib.EgotapError(f"parameter {k}: need a contiguous fp32 tensor (or int64 counter) on {left.device}")
        BB, AB = "backbone.backbone.backbone.", "after_backbone."
        bn = lambda k: (sd[k + ".weight"], sd[k + ".bias"], sd[k + ".running_mean"], sd[k + ".running_var"])       # noqa: E731
        B, S0, dev = left.shape[0], left.shape[2], left.device
        N2 = 2 * B
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)     # noqa: E731
        l0 = new(N2, 64, S0 // 2, S0 // 2)
        H.stem_bn_fwd(left, right, sd[BB + "conv1.weight"], bn(BB + "bn1"), l0)
        x = new(N2, 64, S0 // 4, S0 // 4)
        H.maxpool_fwd(l0, x)
        del l0
        side, pyr = S0 // 4, []
        for i, (c, st) in enumerate(_spec.HM_STAGES, start=1):
            for b in range(self.blocks[i - 1]):
                k = f"{BB}layer{i}.{b}."
                stride = st if b == 0 else 1
                so = side // stride
                t1 = new(N2, c, side, side)
                H.conv_bn_fwd(h, x, sd[k + "conv1.weight"], bn(k + "bn1"), t1, taps=1)
                t2 = new(N2, c, so, so)
                H.conv_bn_fwd(h, t1, sd[k + "conv2.weight"], bn(k + "bn2"), t2, taps=9, stride=stride)
                idt = x
                if (k + "downsample.0.weight") in sd:
                    idt = new(N2, 4 * c, so, so)
                    H.conv_bn_fwd(h, x, sd[k + "downsample.0.weight"], bn(k + "downsample.1"), idt, taps=1, stride=stride, relu=False)
                y = new(N2, 4 * c, so, so)
                H.conv_bn_fwd(h, t2, sd[k + "conv3.weight"], bn(k + "bn3"), y, res=idt, taps=1)
                x, side = y, so
            pyr.append(x)
        L = [t.view(B, 2 * t.shape[1], t.shape[2], t.shape[3]) for t in pyr]
        f = 2 * _spec.hm_feature_scale(self.model_name)
        wb = lambda name: (sd[AB + name + ".weight"], sd[AB + name + ".bias"])      # noqa: E731
        s64, s32, s16, s8 = (t.shape[2] for t in L)
        w, bias = wb("layer4_1x1.0")
        u4 = new(B, 512 * f, s8, s8)
        H.conv_fwd(h, L[3], w, u4, bias=bias, taps=1, relu=True)
        cat3 = new(B, (512 + 258) * f, s16, s16)
        H.upsample_fwd(u4, H.View(cat3, 0, 512 * f))
        w, bias = wb("layer3_1x1.0")
        H.conv_fwd(h, L[2], w, H.View(cat3, 512 * f, 258 * f), bias=bias, taps=1, relu=True)
        x3 = new(B, 512 * f, s16, s16)
        w, bias = wb("conv_up3.0")
        H.conv_fwd(h, cat3, w, x3, bias=bias, taps=9, relu=True)
        del cat3, u4
        cat2 = new(B, (512 + 128) * f, s32, s32)
        H.upsample_fwd(x3, H.View(cat2, 0, 512 * f))
        w, bias = wb("layer2_1x1.0")
        H.conv_fwd(h, L[1], w, H.View(cat2, 512 * f, 128 * f), bias=bias, taps=1, relu=True)
        x2 = new(B, 256 * f, s32, s32)
        w, bias = wb("conv_up2.0")
        H.conv_fwd(h, cat2, w, x2, bias=bias, taps=9, relu=True)
        del cat2, x3
        cat1 = new(B, (256 + 64) * f, s64, s64)
        H.upsample_fwd(x2, H.View(cat1, 0, 256 * f))
        w, bias = wb("layer1_1x1.0")
        H.conv_fwd(h, L[0], w, H.View(cat1, 256 * f, 64 * f), bias=bias, taps=1, relu=True)
        x1 = new(B, 256 * f, s64, s64)
        w, bias = wb("conv_up1.0")
        H.conv_fwd(h, cat1, w, x1, bias=bias, taps=9, relu=True)
        w, bias = wb("conv_heatmap")
        H.conv_fwd(h, x1, w, H.View(out, channel_offset, 2 * self.num_heatmap), bias=bias, taps=1, relu=False)
        return out

    def forward(self, *inputs):
        if len(inputs) != 2:
            raise NotImplementedError("stereo input (left, right) expected")
        if not inputs[0].is_cuda:
            raise _lib.EgotapError("HeatMap_UnrealEgo_Shared runs on the GPU only (no CPU fallback)")
        if self.bottleneck and self.training:
            raise NotImplementedError(f"backbone {self.model_name!r}: the eval forward is built (fp32); train-mode BatchNorm / stage-1 training cover "
                                      "resnet18 and resnet34 -- call .eval() (the stage-2 wrapper keeps frozen estimators in eval mode)")
        if self.training and torch.is_grad_enabled():
            from .hm_training import hm_train_forward          # train mode: batch-statistics BatchNorm2d, differentiable
            return hm_train_forward(self, inputs[0], inputs[1])
        left, right = (t.detach().float().contiguous() for t in inputs)
        out = torch.empty((left.shape[0], 2 * self.num_heatmap, self.hm_size, self.hm_size), dtype=torch.float32,
                          device=left.device)
        return self.forward_into(left, right, out)
