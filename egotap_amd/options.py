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


def parse(argv=None) -> types.SimpleNamespace:
    """argparse front end with the reference's flag names (only the flags the hot path reads)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="egotap_autoencoder")
    ap.add_argument("--model_name", default="resnet18")
    ap.add_argument("--joint_preset", default="UnrealEgo")
    ap.add_argument("--num_heatmap", type=int, default=15)
    ap.add_argument("--num_rot_heatmap", type=int, default=15)
    ap.add_argument("--heatmap_type", default="sin")
    ap.add_argument("--ae_hidden_size", type=int, default=128)
    ap.add_argument("--patched_heatmap_ae", action="store_true")
    ap.add_argument("--skel_layer", default="PU")
    ap.add_argument("--load_size_heatmap", nargs="+", type=int, default=[64, 64])
    ap.add_argument("--use_amp", action="store_true")
    ap.add_argument("--use_gt_heatmap", action="store_true")
    ap.add_argument("--gpu_ids", default="0")
    ap.add_argument("--batch_size", type=int, default=32)
    ns = ap.parse_args(argv)
    opt = preset_defaults(ns.joint_preset, ns.load_size_heatmap[0])
    for k, v in vars(ns).items():
        setattr(opt, k, v)
    opt.gpu_ids = [int(i) for i in str(ns.gpu_ids).split(",") if int(i) >= 0]
    opt.estimate_head = opt.joint_preset == "UnrealEgo"      # options/dataset_options.py:33-38
    opt.stereo = True
    return opt
