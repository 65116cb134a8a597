"""The subset of the reference's command-line options the hot path reads, with the values of the shipped
scripts (scripts/test/unrealego.sh, scripts/test/egocap.sh; options/base_options.py, dataset_options.py).
Flag names are the reference's."""
from __future__ import annotations

import argparse
import types


def preset_defaults(joint_preset: str = "UnrealEgo", hm_size: int = 64) -> types.SimpleNamespace:
    if joint_preset not in ("UnrealEgo", "EgoCap"):
        raise ValueError("joint_preset is {} which is undefined".format(joint_preset))
    nj = 15 if joint_preset == "UnrealEgo" else 17
    return types.SimpleNamespace(
        model="egotap_autoencoder", model_name="resnet18", joint_preset=joint_preset, num_heatmap=nj, num_rot_heatmap=nj,
        heatmap_type="sin", ae_hidden_size=128, patched_heatmap_ae=True, skel_layer="PU", load_size_heatmap=[hm_size, hm_size],
        estimate_head=joint_preset == "UnrealEgo", stereo=True, init_ImageNet=False, init_type="kaiming", use_amp=False,
        use_gt_heatmap=False, gpu_ids=[0], isTrain=False, batch_size=32, path_to_trained_heatmap=None, distributed=False,
        lambda_mpjpe=0.1, lambda_cos_sim=-0.01, log_dir="./log", experiment_name="experiment")


def parse(argv=None, is_train: bool = False) -> types.SimpleNamespace:
    """argparse front end with the reference's flag names and defaults (options/base_options.py:14-82, dataset_options.py:12-27,
    train_options.py:7-88): every flag the shipped scripts pass (scripts/train/**.sh, scripts/test/**.sh) parses, the ones the
    hot path does not read (display, data paths, unused loss weights) are carried on the namespace untouched."""
    ap = argparse.ArgumentParser()
    A = ap.add_argument
    A("--project_name", default="project_name"); A("--experiment_name", default="experiment"); A("--which_epoch", default="latest")
    A("--gpu_ids", default="0"); A("--model", default="egotap_autoencoder"); A("--init_ImageNet", action="store_true")
    A("--model_name", default="resnet18"); A("--use_slurm", action="store_true"); A("--use_amp", action="store_true")
    A("--path_to_trained_heatmap", default=None); A("--path_to_trained_decoder", default=None); A("--use_init_pose", action="store_true")
    A("--skel_layer", default="LSTM"); A("--patched_heatmap_ae", action="store_true"); A("--use_gt_heatmap", action="store_true")
    A("--num_heatmap", type=int, default=15); A("--num_rot_heatmap", type=int, default=0); A("--heatmap_type", default="none")
    A("--num_imu", type=int, default=5); A("--num_threads", type=int, default=8); A("--batch_size", type=int, default=16)
    A("--load_size_heatmap", nargs="+", type=int, default=[64, 64]); A("--ae_hidden_size", type=int, default=20)
    A("--init_type", default="kaiming"); A("--experiment", action="store_true"); A("--distributed", action="store_true")
    A("--log_dir", default="./log")
    A("--default_data_path", default="./UnrealEgoData"); A("--data_dir", default=""); A("--data_sub_path", default="")
    A("--metadata_dir", nargs="+", default=[]); A("--data_prefix", default=""); A("--joint_preset", default="UnrealEgo")
    if is_train:
        A("--epoch_count", type=int, default=1); A("--niter", type=int, default=0); A("--niter_decay", type=int, default=0)
        A("--continue_train", action="store_true"); A("--optimizer_type", default="Adam"); A("--lr_policy", default="lambda")
        A("--lr_decay_iters_step", type=int, default=4); A("--lr", type=float, default=1e-3); A("--weight_decay", type=float, default=0.0)
        A("--opt_eps", type=float, default=1e-4); A("--decouple", action="store_true")
        for name, dflt in (("mpjpe", 1.0), ("pelvis", 0.01), ("rot", 1.0), ("heatmap", 1.0), ("segmentation", 1.0), ("rot_heatmap", 1.0),
                           ("pose", 0.1), ("indep_pos", 0.1), ("heatmap_rec", 1e-3), ("rot_heatmap_rec", 1e-3), ("cos_sim", -1e-2)):
            A("--lambda_" + name, type=float, default=dflt)
        A("--stage", action="append", dest="train_stage", default=[]); A("--auto_restart", action="store_true")
        A("--auto_terminate", action="store_true")
        # additions of this build (not reference flags), see INTEGRATION.md: reduced-precision mode behind --use_amp; opt-OUT of the
        # reference's batch-statistics BatchNorm in the frozen estimators (train.py:91: the default here, as there)
        A("--amp_precision", default="bf16", choices=["bf16", "bf16x3"]); A("--frozen_heatmap_bn_eval", action="store_true")
    ns = ap.parse_args(argv)
    opt = preset_defaults(ns.joint_preset, ns.load_size_heatmap[0])
    for k, v in vars(ns).items():
        setattr(opt, k, v)
    opt.gpu_ids = [int(i) for i in str(ns.gpu_ids).split(",") if int(i) >= 0]
    opt.estimate_head = opt.joint_preset == "UnrealEgo"      # options/dataset_options.py:33-38
    opt.stereo = True
    opt.isTrain = is_train
    if not is_train:
        opt.use_amp = False                                    # options/test_options.py:15
    return opt


def parse_train(argv=None) -> types.SimpleNamespace:
    """TrainOptions().parse() of the reference (options/train_options.py): isTrain = True"""
    return parse(argv, is_train=True)
