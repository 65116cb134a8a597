"""Host-side mirror of the reference's model wrapper for the hot path.

``create_model(opt)`` / ``EgoTAPAutoEncoderModel`` keep the surface ``test.py`` / ``utils/evaluate.py`` touch
(reference: model/models.py:2-17, model/egotap_autoencoder_model.py:13-350, model/base_model.py): ``set_input``,
``forward(evaluate=)``, ``evaluate(dict)``, ``set_eval_mode``, ``eval_key``, ``load_networks`` / ``save_networks``,
the ``pred_*`` attributes, ``optimize_parameters`` / ``update_learning_rate`` for training, and ``HeatmapSharedModel`` for
stage-1 training of one heatmap estimator (model/heatmap_shared_model.py).  The networks run through libegotap_hip.so; this file
is plumbing only.
"""
from __future__ import annotations

import copy
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import networks


def create_model(opt):
    name = getattr(opt, "model", "egotap_autoencoder")
    if name == "egotap_autoencoder":
        model = EgoTAPAutoEncoderModel()
    elif name == "heatmap_shared":
        model = HeatmapSharedModel()
    else:
        raise ValueError("Model [%s] not recognized." % name)
    model.initialize(opt)
    return model


class EgoTAPAutoEncoderModel(nn.Module):
    def name(self):
        return "EgoTAP AutoEncoder model"

    def initialize(self, opt):
        self.opt = opt
        self.gpu_ids = getattr(opt, "gpu_ids", [0])
        self.isTrain = getattr(opt, "isTrain", False)
        self.save_dir = os.path.join(getattr(opt, "log_dir", "./log"), getattr(opt, "experiment_name", "experiment"))
        self.device = torch.device("cuda:{}".format(self.gpu_ids[0])) if self.gpu_ids else torch.device("cuda:0")
        self.loss_names = ["pose", "cos_sim"] if self.isTrain else []
        self.model_names = ["HeatMap", "RotHeatMap", "AutoEncoder"]
        self.visual_names, self.visual_pose_names = [], ["pred_pose", "gt_pose"]
        self.eval_key = "mpjpe"
        self.cm2mm = 10
        self.stereo = getattr(opt, "stereo", True)
        self.input_channel_scale = 2 if self.stereo else 1
        if not self.stereo:
            raise NotImplementedError("only the stereo presets are built")
        pos_opt, rot_opt = copy.deepcopy(opt), copy.deepcopy(opt)      # egotap_autoencoder_model.py:104-107
        pos_opt.num_rot_heatmap = 0
        rot_opt.num_heatmap = 0
        self.net_HeatMap = networks.HeatMap_UnrealEgo_Shared(pos_opt, getattr(opt, "model_name", "resnet18"), 2)
        self.net_RotHeatMap = networks.HeatMap_UnrealEgo_Shared(rot_opt, getattr(opt, "model_name", "resnet18"), 2)
        self.net_AutoEncoder = networks.EgoTAPAutoEncoder(opt, input_channel_scale=2)
        self.optimizers, self.schedulers = [], []
        self.to(self.device)
        self._hm_ws = None
        # --use_amp (egotap_autoencoder_model.py:21, 219, 317-323: fp16 autocast + GradScaler around the training forward / loss):
        # mapped to the reduced-precision HIP mode -- bf16 matrix-core arithmetic with fp32 accumulation and fp32 master weights.
        # bf16 keeps fp32's exponent range, so there is no loss scaling and no scaler state; evaluation stays fp32, as the
        # reference disables autocast there (options/test_options.py:15, forward(evaluate=True)).
        self.use_amp = bool(getattr(opt, "use_amp", False)) and self.isTrain
        self.amp_precision = getattr(opt, "amp_precision", "bf16")
        if self.isTrain:
            path = getattr(opt, "path_to_trained_heatmap", None)
            if path is not None:
                # egotap_autoencoder_model.py:113-126: <dir>_pos/<file> -> net_HeatMap, <dir>_<heatmap_type>/<file> -> net_RotHeatMap
                d, f = os.path.dirname(path), os.path.basename(path)
                self.load_networks(net=self.net_HeatMap, path_to_trained_weights=os.path.join(d + "_pos", f))
                self.load_networks(net=self.net_RotHeatMap,
                                   path_to_trained_weights=os.path.join(d + "_" + getattr(opt, "heatmap_type", "sin"), f))
            elif not getattr(opt, "use_gt_heatmap", False):
                # the reference would train from RGB through never-trained estimators whose parameters are not even in the
                # optimizer (egotap_autoencoder_model.py:144-148): lifting from noise.  Refuse instead of doing that silently.
                raise ValueError("training the lifting head from RGB needs --path_to_trained_heatmap (stage-1 checkpoints "
                                 "<dir>_pos/<file> and <dir>_<heatmap_type>/<file>), or --use_gt_heatmap")
            # heatmap estimators are frozen while the lifting head trains (egotap_autoencoder_model.py:127-129, 144-148)
            for n in (self.net_HeatMap, self.net_RotHeatMap):
                for prm in n.parameters():
                    prm.requires_grad = False
            from .training import EgotapAdamW
            if getattr(opt, "optimizer_type", "AdamW") != "AdamW":
                raise NotImplementedError("only AdamW (the shipped training scripts) is built")
            self.optimizer_AutoEncoder = EgotapAdamW(self.net_AutoEncoder.parameters(), lr=getattr(opt, "lr", 1e-3),
                                                     eps=getattr(opt, "opt_eps", 1e-4), weight_decay=getattr(opt, "weight_decay", 0.0))
            self.optimizers.append(self.optimizer_AutoEncoder)
            if getattr(opt, "lr_policy", None):                      # egotap_autoencoder_model.py:151-152
                from .training import get_scheduler
                self.schedulers = [get_scheduler(o, opt) for o in self.optimizers]

    # ---- data --------------------------------------------------------------------------------------------------
    def set_input(self, data):
        """Keys of dataloader/data_loader.py:166-215; only the ones the eval path reads are required."""
        self.data = data
        dev = self.device
        self.input_rgb_left = data["input_rgb_left"].to(dev, non_blocking=True)
        self.input_rgb_right = data["input_rgb_right"].to(dev, non_blocking=True)
        for k in ("gt_heatmap_left", "gt_heatmap_right", "gt_limb_heatmap_left", "gt_limb_heatmap_right"):
            setattr(self, k, data[k].to(dev, non_blocking=True) if k in data else None)
        self.gt_pose = data["gt_local_pose"].to(dev, non_blocking=True) if "gt_local_pose" in data else None
        if self.gt_heatmap_left is None and "gt_camera_2d_left" in data and getattr(self.opt, "use_gt_heatmap", False):
            # joints instead of rendered heatmaps: synthesise them on the device (the data loader's per-frame CPU work,
            # dataloader/data_loader.py:76-215); gt_local_pose_full = all J+1 joints incl. the root, for the limb directions
            from . import lib as _lib
            full = data.get("gt_local_pose_full", data["gt_local_pose"]).to(dev)
            syn = _lib.synth_heatmaps(data["gt_camera_2d_left"].to(dev), data["gt_camera_2d_right"].to(dev), full,
                                      self.opt.joint_preset, self.net_AutoEncoder.preset.hm_size)
            for k in ("gt_heatmap_left", "gt_heatmap_right", "gt_limb_heatmap_left", "gt_limb_heatmap_right"):
                setattr(self, k, syn[k])
            self._gt_cat = syn["cat"]

    # ---- forward -----------------------------------------------------------------------------------------------
    def _estimator_bn_modes(self):
        """(position net, limb net) -> True where the estimator normalises with BATCH statistics.  The reference's estimators are plain
        sub-modules of the wrapper: they run in whatever mode their `.training` flag says.  train.py:91 `model.train()` therefore
        leaves the FROZEN estimators' BatchNorm2d on batch statistics, running statistics drifting, while the head trains
        (egotap_autoencoder_model.py:127-129 freezes parameters only), and set_eval_mode() (:325-327) switches net_HeatMap but not
        net_RotHeatMap.  That is the default here too (pinned by tests/golden/wrapper_step_rgb_ue_b2.npz, the reference wrapper's own
        run).  opt.frozen_heatmap_bn_eval = True is the opt-out: folded running-statistics BatchNorm in both estimators whatever
        their mode (what the stage-1 checkpoints were validated with; frames then stay independent and large batches run in chunks on
        the bf16 channels-last kernels)."""
        if getattr(self.opt, "frozen_heatmap_bn_eval", False):
            return False, False
        return bool(self.net_HeatMap.training), bool(self.net_RotHeatMap.training)

    @staticmethod
    def _adjacent_channel_slices(parts):
        """True when the tensors are consecutive dim-1 slices of ONE contiguous fp32 CUDA [B, C, S, S] tensor that holds nothing else per frame"""
        a = parts[0]
        if not all(t is not None and t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 for t in parts):
            return False
        S2 = a.shape[2] * a.shape[3]
        ctot = sum(t.shape[1] for t in parts)
        want_stride = (ctot * S2, S2, a.shape[3], 1)
        off = a.storage_offset()
        base = a.untyped_storage().data_ptr()
        for t in parts:
            if t.untyped_storage().data_ptr() != base or tuple(t.stride()) != want_stride or t.shape[0] != a.shape[0] or tuple(t.shape[2:]) != tuple(a.shape[2:]) \
                    or t.storage_offset() != off:
                return False
            off += t.shape[1] * S2
        return True

    def forward_heatmap(self):
        p = self.net_AutoEncoder.preset
        J = p.n_joints_hm
        if getattr(self.opt, "use_gt_heatmap", False):
            parts = (self.gt_heatmap_left, self.gt_heatmap_right, self.gt_limb_heatmap_left, self.gt_limb_heatmap_right)
            if getattr(self, "_gt_cat", None) is not None and self.gt_heatmap_left.data_ptr() == self._gt_cat.data_ptr():
                cat = self._gt_cat                     # synthesised in place in the head's layout: no torch.cat
            elif self._adjacent_channel_slices(parts):
                # [r5] the four maps already ARE consecutive channel slices of one fp32 [B, 6J, S, S] tensor (a loader that renders into the
                # head's layout, bench.py's resident inputs): the concatenation of egotap_autoencoder_model.py:195-216 is that tensor -- a view,
                # not a 1.4 GB copy per 1024-frame step
                a = parts[0]
                cat = a.as_strided((a.shape[0], sum(t.shape[1] for t in parts), a.shape[2], a.shape[3]), a.stride(), a.storage_offset())
            else:
                cat = torch.cat(parts, dim=1).float().contiguous()
        else:
            left = self.input_rgb_left.float().contiguous()
            right = self.input_rgb_right.float().contiguous()
            B = left.shape[0]
            cat = torch.empty((B, p.in_channels, p.hm_size, p.hm_size), dtype=torch.float32, device=left.device)
            bn_batch = self._estimator_bn_modes()
            # position net: channels [0, 2J) (left | right), limb net: [2J, 6J) (left cos, sin | right cos, sin)
            for net, c0, cn, batch_stats in ((self.net_HeatMap, 0, 2 * J, bn_batch[0]), (self.net_RotHeatMap, 2 * J, 4 * J, bn_batch[1])):
                if batch_stats and getattr(net, "bottleneck", False):
                    # resnet50 / resnet101 estimators have no batch-statistics forward: keep the reference command line running on the
                    # eval-mode (folded running statistics) forward below and say so once -- the deviation is exactly what
                    # --frozen_heatmap_bn_eval selects explicitly
                    if not getattr(self, "_warned_bottleneck_bn", False):
                        import warnings
                        warnings.warn(f"frozen {net.model_name} estimators run with running-statistics BatchNorm although the wrapper is in train mode "
                                      "(train.py:91 would use batch statistics; only resnet18 / resnet34 have that forward here): same as "
                                      "--frozen_heatmap_bn_eval; running statistics are not updated", RuntimeWarning, stacklevel=3)
                        self._warned_bottleneck_bn = True
                    batch_stats = False
                if batch_stats and getattr(net, "precision", "f32") == "bf16" and B >= 2:
                    # [r5] --use_amp: batch-statistics BatchNorm on the bf16 channels-last kernels (egotap_hm_forward_bnbatch) -- the backbone over
                    # the whole batch (the statistics couple its frames), the BatchNorm-free decoder in hm_chunk pieces, straight into the
                    # head's input slice; one scratch shared by both estimators
                    chunk = min(B, int(getattr(self.opt, "hm_chunk", 256)))
                    net.forward_bnbatch_into(left, right, cat, c0, chunk=chunk, workspace=self.net_HeatMap.bnbatch_workspace(B, chunk, left.device))
                elif batch_stats:
                    # fp32 / bf16x3: the stage-1 train-mode forward without a graph (conv_f32 / conv_bf16 kernels + bn2d_fwd), whole batch
                    from .hm_training import hm_train_forward_nograd
                    cat[:, c0:c0 + cn] = hm_train_forward_nograd(net, left, right)
                else:
                    # eval-mode estimators treat frames independently: walk a large batch in chunks so that the U-Net scratch stays
                    # at the chunk's size (B = 1024 from RGB, BASELINE config 3); one scratch shared by both estimators
                    was = net.training
                    net.eval()
                    try:
                        chunk = min(B, int(getattr(self.opt, "hm_chunk", 256)))
                        ws = None if net.bottleneck else self.net_HeatMap._workspace(chunk, left.device)   # (Bottleneck nets allocate per call)
                        for lo in range(0, B, chunk):
                            hi = min(B, lo + chunk)
                            net.forward_into(left[lo:hi], right[lo:hi], cat[lo:hi], c0, workspace=ws)
                    finally:
                        net.train(was)
        self.pred_heatmap_cat = cat
        self.pred_heatmap_left, self.pred_heatmap_right = cat[:, :J], cat[:, J:2 * J]
        self.pred_limb_heatmap_left, self.pred_limb_heatmap_right = cat[:, 2 * J:4 * J], cat[:, 4 * J:]

    def forward(self, evaluate=False):
        with torch.no_grad():                      # the estimators never train here (egotap_autoencoder_model.py:179 with train_heatmap False)
            self.forward_heatmap()
        if self.net_AutoEncoder.training:          # (evaluate only switches autocast off in the reference: egotap_autoencoder_model.py:219)
            from .training import lift_train_forward
            self.pred_pose = lift_train_forward(self.net_AutoEncoder, self.pred_heatmap_cat)
            _, self.pred_rot, self.pred_indep_pos, rec = self.net_AutoEncoder._zero_outputs(self.pred_heatmap_cat.shape[0], self.pred_pose.device)
        else:
            self.pred_pose, self.pred_rot, self.pred_indep_pos, rec = self.net_AutoEncoder(
                self.pred_heatmap_cat, self.input_rgb_left, self.input_rgb_right)
        J = self.net_AutoEncoder.preset.n_joints_hm
        self.pred_heatmap_rec_cat = rec
        self.pred_heatmap_left_rec, self.pred_heatmap_right_rec = rec[:, :J], rec[:, J:2 * J]
        self.pred_limb_heatmap_left_rec, self.pred_limb_heatmap_right_rec = rec[:, 2 * J:4 * J], rec[:, 4 * J:]

    def backward_AutoEncoder(self):
        from .training import PoseLossFn
        lam_m = getattr(self.opt, "lambda_mpjpe", 0.1)
        lam_c = getattr(self.opt, "lambda_cos_sim", -0.01)
        both = PoseLossFn.apply(self.net_AutoEncoder, self.pred_pose, self.gt_pose, lam_m, lam_c)
        self.loss_pose, self.loss_cos_sim = both[0], both[1]
        self.loss_total = self.loss_total + both.sum()

    def optimize_parameters(self):
        """One step of egotap_autoencoder_model.py:299-323: forward, loss, backward, AdamW -- all on HIP kernels (fp32, or the
        bf16 mode under --use_amp; no GradScaler: bf16 has fp32's exponent range)."""
        if not self.isTrain:
            raise RuntimeError("optimize_parameters() needs a model created with opt.isTrain = True")
        self.net_AutoEncoder.train()
        if self.use_amp:           # --use_amp: reduced-precision training arithmetic (opt.amp_precision, default "bf16"); the reference's
            # autocast spans the frozen estimators' forward too (egotap_autoencoder_model.py:219).  The requested mode is set
            # explicitly: whatever a caller (or evaluate()) left on the networks, the step runs in opt.amp_precision.
            # The mode is opt.amp_precision -- unless the caller chose a reduced mode for the whole model explicitly with
            # model.set_precision("bf16x3" | "bf16"): that choice wins over the flag's default and is kept from step to step.
            want = getattr(self, "_explicit_precision", None) or self.amp_precision
            for n in (self.net_AutoEncoder, self.net_HeatMap, self.net_RotHeatMap):
                if getattr(n, "bottleneck", False):
                    continue                                     # resnet50 / resnet101 estimators run in fp32 only (frozen here anyway)
                if getattr(n, "precision", "f32") != want:
                    n.set_precision(want)
        for o in self.optimizers:
            o.zero_grad()
        self.forward()
        self.loss_total = 0.0
        self.backward_AutoEncoder()
        self.loss_total.backward()          # data parallel: the gradient all-reduce runs INSIDE the backward, bucket by bucket, overlapped
        for o in self.optimizers:          # with it (parallel.GradReducer on the flat gradient arena); the gradients arrive averaged
            o.step()

    def set_precision(self, mode: str = "f32"):
        """f32 (default) | bf16x3 | bf16 for the three networks (the reference's analogous switch is --use_amp).
        Under --use_amp the training step runs in opt.amp_precision; an explicit set_precision("bf16x3" | "bf16") on the model overrides
        that default for every following step, set_precision("f32") hands the choice back to the flag (a training step under --use_amp
        is never fp32: the reference's is not either)."""
        self._explicit_precision = mode if mode != "f32" else None
        for n in (self.net_HeatMap, self.net_RotHeatMap, self.net_AutoEncoder):
            if getattr(n, "bottleneck", False) and mode != "f32":
                continue                                         # resnet50 / resnet101 estimators: fp32 only
            n.set_precision(mode)
        return self

    def set_eval_mode(self):
        """egotap_autoencoder_model.py:325-327: the head and the POSITION estimator; net_RotHeatMap keeps its mode, exactly as in the
        reference (every caller there runs model.eval() first: utils/evaluate.py:93, 150).  opt.frozen_heatmap_bn_eval makes the
        estimators' mode irrelevant."""
        self.net_AutoEncoder.eval()
        self.net_HeatMap.eval()

    def evaluate(self, runnning_average_dict):
        self.set_eval_mode()
        with torch.no_grad():
            nets = (self.net_AutoEncoder, self.net_HeatMap, self.net_RotHeatMap)
            prec = [getattr(n, "precision", "f32") for n in nets]
            try:
                if self.use_amp:
                    for n, q in zip(nets, prec):
                        if q != "f32":
                            n.set_precision("f32")               # autocast is off in evaluation (forward(evaluate=True))
                self.forward(evaluate=True)
            finally:                                             # an exception in the forward must not leave training in fp32
                if self.use_amp:
                    for n, q in zip(nets, prec):
                        if getattr(n, "precision", "f32") != q:
                            n.set_precision(q)
            from . import lib as _lib                     # one fused launch: per-sample MPJPE + Procrustes-aligned MPJPE
            # batches of 2 or 3 frames: the reference's batch_compute_similarity_transform_torch aligns the wrong axes there
            # (utils/util.py:337) and test.py prints that number; reproduced by default, opt.pa_mpjpe_reference_batch_axes = False
            # gives every frame the PA-MPJPE it has in any other batch
            err, pa = _lib.pose_metrics(self.pred_pose, self.gt_pose,
                                        reference_batch_axes=bool(getattr(self.opt, "pa_mpjpe_reference_batch_axes", True)))
            err, pa = (err * self.cm2mm).cpu(), (pa * self.cm2mm).cpu()      # one device->host copy, not one per sample
        for i in range(self.pred_pose.shape[0]):
            runnning_average_dict.update(dict(mpjpe=err[i], pa_mpjpe=pa[i]))
        return self.pred_pose, self.pred_heatmap_cat, runnning_average_dict

    # ---- checkpoints (base_model.py:64-148 file naming) --------------------------------------------------------
    def save_networks(self, which_epoch=None, checkpoint_path=None):
        if which_epoch is None and checkpoint_path is None:
            raise ValueError("which_epoch and checkpoint_path cannot be both None")
        which_epoch = "checkpoint" if which_epoch is None else which_epoch
        checkpoint_path = self.save_dir if checkpoint_path is None else checkpoint_path
        os.makedirs(checkpoint_path, exist_ok=True)
        for name in self.model_names:
            net = getattr(self, "net_" + name)
            sd = OrderedDict((k, v.detach().cpu()) for k, v in net.state_dict().items())
            torch.save(sd, os.path.join(checkpoint_path, "%s_net_%s.pth" % (which_epoch, name)))
        for i, o in enumerate(self.optimizers):                      # base_model.py:83-92
            torch.save(o.state_dict(), os.path.join(checkpoint_path, "%s_optim_%s.pth" % (which_epoch, i)))
        for i, sch in enumerate(self.schedulers):
            torch.save(sch.state_dict(), os.path.join(checkpoint_path, "%s_scheduler_%s.pth" % (which_epoch, i)))
        if isinstance(which_epoch, int) and which_epoch > 1 and which_epoch != getattr(self.opt, "epoch_count", None):
            prev = which_epoch - 1                                   # base_model.py:94-114: keep only the latest numbered epoch
            names = ["%s_net_%s.pth" % (prev, n) for n in self.model_names]
            names += ["%s_optim_%s.pth" % (prev, i) for i in range(len(self.optimizers))]
            names += ["%s_scheduler_%s.pth" % (prev, i) for i in range(len(self.schedulers))]
            for fn in names:
                fp = os.path.join(checkpoint_path, fn)
                if os.path.exists(fp):
                    os.remove(fp)

    def load_optimizers(self, which_epoch="checkpoint", checkpoint_path=None):
        """resume: optimizer / scheduler state written by save_networks (also accepts torch.optim.AdamW state files)"""
        checkpoint_path = self.save_dir if checkpoint_path is None else checkpoint_path
        for i, o in enumerate(self.optimizers):
            o.load_state_dict(torch.load(os.path.join(checkpoint_path, "%s_optim_%s.pth" % (which_epoch, i)), map_location=self.device))
        for i, sch in enumerate(self.schedulers):
            sch.load_state_dict(torch.load(os.path.join(checkpoint_path, "%s_scheduler_%s.pth" % (which_epoch, i))))

    def load_networks(self, which_epoch=None, net=None, path_to_trained_weights=None, checkpoint_path=None):
        if path_to_trained_weights is not None:
            # base_model.py:136-146: "./log/" is stripped and the rest joined with opt.log_dir; module. prefixes of a
            # DataParallel checkpoint are dropped under --distributed (here: always, it is harmless)
            if "./log" in path_to_trained_weights:
                path_to_trained_weights = path_to_trained_weights.replace("./log/", "")
            weight_path = os.path.join(getattr(self.opt, "log_dir", "./log"), path_to_trained_weights)
            sd = torch.load(weight_path, map_location="cpu")
            sd = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in sd.items())
            net.load_state_dict(sd)
            return
        if which_epoch is None and checkpoint_path is None:
            raise ValueError("which_epoch and checkpoint_path cannot be both None")
        which_epoch = "checkpoint" if which_epoch is None else which_epoch
        checkpoint_path = self.save_dir if checkpoint_path is None else checkpoint_path
        for name in self.model_names:
            n = getattr(self, "net_" + name)
            n.load_state_dict(torch.load(os.path.join(checkpoint_path, "%s_net_%s.pth" % (which_epoch, name)), map_location="cpu"))
            n.eval() if not self.isTrain else n.train()

    def get_current_errors(self):
        return OrderedDict((n, getattr(self, "loss_" + n).item()) for n in self.loss_names if hasattr(self, "loss_" + n))

    def update_learning_rate(self):
        for s in self.schedulers:
            s.step()


class HeatmapSharedModel(nn.Module):
    """Stage-1 wrapper: trains / evaluates one heatmap estimator (reference: model/heatmap_shared_model.py).  Same surface as the
    reference (set_input keys, forward, optimize_parameters, evaluate -> mse_heatmap, loss_* attributes, save / load); the network,
    the losses and the optimizer step run on HIP kernels (egotap_amd/hm_training.py), PyTorch is autograd glue only.
    Built for the two shipped configurations: the position net (num_rot_heatmap = 0) or the sin/cos limb net (num_heatmap = 0)."""

    def name(self):
        return "Heatmap Shared model"

    def initialize(self, opt):
        self.opt = opt
        self.gpu_ids = getattr(opt, "gpu_ids", [0])
        self.isTrain = getattr(opt, "isTrain", False)
        self.save_dir = os.path.join(getattr(opt, "log_dir", "./log"), getattr(opt, "experiment_name", "experiment"))
        self.device = torch.device("cuda:{}".format(self.gpu_ids[0])) if self.gpu_ids else torch.device("cuda:0")
        if not getattr(opt, "stereo", True):
            raise NotImplementedError("only the stereo presets are built")
        if opt.num_heatmap > 0 and opt.num_rot_heatmap > 0:
            raise NotImplementedError("train the position net and the limb net separately (the shipped stage-1 scripts do)")
        self.is_limb = opt.num_rot_heatmap > 0
        self.loss_names = ["limb_heatmap_left", "limb_heatmap_right"] if self.is_limb else ["heatmap_left", "heatmap_right"]
        self.model_names = ["HeatMap"]
        self.visual_names, self.visual_pose_names = ["input_rgb_left", "input_rgb_right"], []
        self.eval_key, self.cm2mm = "mse_heatmap", 10
        self.net_HeatMap = networks.HeatMap_UnrealEgo_Shared(opt, getattr(opt, "model_name", "resnet18"), 2)
        if self.isTrain and self.net_HeatMap.bottleneck:
            raise NotImplementedError(f"stage-1 training of a {self.net_HeatMap.model_name} estimator is not built (resnet18 / resnet34 are); "
                                      "its checkpoints load and run in evaluation and as frozen estimators of stage 2")
        self.optimizers, self.schedulers = [], []
        self.to(self.device)
        # --use_amp (heatmap_shared_model.py:17, 99, 111: fp16 autocast + GradScaler): the reduced-precision HIP mode of the
        # estimator's convolutions (split-bf16 matrix-core products, fp32 accumulate, fp32 master weights; no scaler needed)
        if self.isTrain and getattr(opt, "use_amp", False):
            self.net_HeatMap.set_precision(getattr(opt, "amp_precision_heatmap", "bf16x3"))
        if self.isTrain and getattr(opt, "path_to_trained_heatmap", None) is not None:
            self.load_networks(net=self.net_HeatMap, path_to_trained_weights=opt.path_to_trained_heatmap)   # heatmap_shared_model.py:60-64
        if self.isTrain:
            if getattr(opt, "weight_decay", 0.0) != 0.0:
                raise NotImplementedError("torch.optim.Adam's L2 weight decay is not built (the shipped scripts use 0)")
            from .training import EgotapAdamW, get_scheduler
            # torch.optim.Adam(lr, weight_decay=0) == AdamW with zero decay: same kernel (heatmap_shared_model.py:69-73, eps 1e-8)
            self.optimizer_HeatMap = EgotapAdamW(self.net_HeatMap.parameters(), lr=getattr(opt, "lr", 1e-3), eps=1e-8, weight_decay=0.0)
            self.optimizers.append(self.optimizer_HeatMap)
            if getattr(opt, "lr_policy", None):
                self.schedulers = [get_scheduler(o, opt) for o in self.optimizers]

    def set_input(self, data):
        self.data = data
        dev = self.device
        self.input_rgb_left = data["input_rgb_left"].to(dev).float().contiguous()
        self.input_rgb_right = data["input_rgb_right"].to(dev).float().contiguous()
        if self.is_limb:
            self.gt_limb_heatmap_left, self.gt_limb_heatmap_right = data["gt_limb_heatmap_left"].to(dev), data["gt_limb_heatmap_right"].to(dev)
            self.gt_plength_left, self.gt_plength_right = data["gt_plength_left"].to(dev), data["gt_plength_right"].to(dev)
            self._gt = torch.cat((self.gt_limb_heatmap_left, self.gt_limb_heatmap_right), 1).float().contiguous()
            self._plen = torch.cat((self.gt_plength_left, self.gt_plength_right), 1).float().contiguous()
        else:
            self.gt_heatmap_left, self.gt_heatmap_right = data["gt_heatmap_left"].to(dev), data["gt_heatmap_right"].to(dev)
            self._gt = torch.cat((self.gt_heatmap_left, self.gt_heatmap_right), 1).float().contiguous()
            self._plen = None

    def forward(self):
        self.pred_heatmap_cat = self.net_HeatMap(self.input_rgb_left, self.input_rgb_right)
        n = self.pred_heatmap_cat.shape[1] // 2
        left, right = self.pred_heatmap_cat[:, :n], self.pred_heatmap_cat[:, n:]
        if self.is_limb:
            self.pred_limb_heatmap_left, self.pred_limb_heatmap_right = left, right
        else:
            self.pred_heatmap_left, self.pred_heatmap_right = left, right

    def backward_HeatMap(self):
        """lambda * (MSE(left) + MSE(right)); limb maps are divided by sqrt(gt_plength) first (heatmap_shared_model.py:109-151)"""
        from . import hm_ops as H
        lam = getattr(self.opt, "lambda_rot_heatmap" if self.is_limb else "lambda_heatmap", 1.0)
        pred = self.pred_heatmap_cat
        gl, dl = H.mse_halves(pred.detach().contiguous(), self._gt, self._plen, lam)
        a, b = self.loss_names
        setattr(self, "loss_" + a, gl[0])
        setattr(self, "loss_" + b, gl[1])
        pred.backward(dl)

    def optimize_parameters(self):
        if not self.isTrain:
            raise RuntimeError("optimize_parameters() needs a model created with opt.isTrain = True")
        self.net_HeatMap.train()
        self.optimizer_HeatMap.zero_grad()
        self.forward()
        self.backward_HeatMap()            # data parallel: the gradient all-reduce runs INSIDE the backward, bucket by bucket on the flat
        self.optimizer_HeatMap.step()      # gradient arena (hm_training.HmTrainFn + parallel.GradReducer), as the lifting head's

    def evaluate(self, runnning_average_dict):
        from . import hm_ops as H
        self.net_HeatMap.eval()
        with torch.no_grad():
            self.forward()
            pred = self.pred_heatmap_cat.contiguous()
            for i in range(pred.shape[0]):                      # per-sample metric, as the reference's loop (:174-217)
                gl, _ = H.mse_halves(pred[i:i + 1], self._gt[i:i + 1], self._plen[i:i + 1] if self._plen is not None else None, 1.0)
                runnning_average_dict.update(dict(mse_heatmap=gl[0] + gl[1]))
        return None, self.pred_heatmap_cat, runnning_average_dict

    def set_eval_mode(self):
        self.net_HeatMap.eval()

    save_networks = EgoTAPAutoEncoderModel.save_networks
    load_networks = EgoTAPAutoEncoderModel.load_networks
    load_optimizers = EgoTAPAutoEncoderModel.load_optimizers
    get_current_errors = EgoTAPAutoEncoderModel.get_current_errors
    update_learning_rate = EgoTAPAutoEncoderModel.update_learning_rate
