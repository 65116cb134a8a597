"""Pins oracle/hm_ref.py (heatmap estimator restatement) against golden vectors from the reference's
HeatMap_UnrealEgo_Shared run over our ResNet-18 stand-in (tools/make_golden.py gen_hm).  CPU only.
The ResNet-18 arithmetic itself is third-party (torchvision, absent): parity of the backbone is unpinned."""
import os

import numpy as np
import pytest
import torch

from egotap_amd import spec
from egotap_amd.synthetic import synth_hm_state_dict, synth_input
from oracle import hm_ref as H

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("tag,n_hm", [("pos", 15), ("rot", 30)])
def test_hm_spec_and_forward(tag, n_hm):
    g = np.load(os.path.join(GOLD, f"hm_full_{tag}.npz"))
    entries = spec.hm_state_spec(n_hm)
    assert [k for k, _, _ in entries] == list(g["state_keys"])
    assert ["x".join(str(d) for d in s) for _, s, _ in entries] == list(g["state_shapes"])
    params = [k for k, _, a in entries if a is None and not spec.is_buffer(k)]
    assert params == list(g["param_keys"])           # named_parameters() de-duplicates the aliased tensors
    sd = H.to_torch_sd(synth_hm_state_dict(n_hm, f"hm_{tag}."))
    left = torch.from_numpy(synth_input("rgb_left", (1, 3, 256, 256), -2.0, 2.0))
    right = torch.from_numpy(synth_input("rgb_right", (1, 3, 256, 256), -2.0, 2.0))
    trace = {}
    with torch.no_grad():
        y = H.hm_forward(left, right, sd, trace)
    assert tuple(y.shape) == tuple(g["out_shape"])
    np.testing.assert_allclose(y[0, 0].numpy(), g["out_ch0"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(y[0, -1].numpy(), g["out_last"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(y.reshape(-1)[::97].numpy(), g["out_sample"], atol=2e-4, rtol=1e-4)
    for k in ("conv_up3", "conv_up2", "conv_up1"):
        np.testing.assert_allclose(trace[k].reshape(-1)[::997].numpy(), g[k + "_sample"], atol=2e-4, rtol=1e-4, err_msg=k)
    for i in range(5):
        np.testing.assert_allclose(trace[f"pyr{i}"].reshape(-1)[::997].numpy(), g[f"pyr{i}_sample"], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("tag,nh,nr", [("pos", 15, 0), ("rot", 0, 15)])
def test_oracle_hm_train_step_matches_reference_wrapper(tag, nh, nr):
    """oracle/hm_ref.hm_train_step (train-mode forward with per-eye batch statistics, MSE losses, autograd) against ONE
    optimize_parameters() of the reference's own HeatmapSharedModel (tools/make_golden.py gen_hm_train): prediction, the two
    loss terms, every gradient (strided sample + norm) and the BatchNorm running statistics after the left and right passes."""
    import os
    from egotap_amd.synthetic import synth_hm_state_dict, synth_input
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"hm_train_step_{tag}.npz"))
    C = nh + 2 * nr
    sd_np = synth_hm_state_dict(C, f"hm_{tag}.")
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items() if v.dtype != np.int64}
    for k, v in sd.items():
        if not (k.endswith("running_mean") or k.endswith("running_var")):
            v.requires_grad_(True)
    B = 2
    left = torch.from_numpy(synth_input(f"tr_rgbL_{tag}", (B, 3, 256, 256), -2.0, 2.0))
    right = torch.from_numpy(synth_input(f"tr_rgbR_{tag}", (B, 3, 256, 256), -2.0, 2.0))
    gt = torch.from_numpy(synth_input(f"tr_gt_{tag}", (B, 2 * C, 64, 64), 0.0, 1.0))
    plen = torch.from_numpy(synth_input(f"tr_plen_{tag}", (B, 2 * C), 2.0, 40.0)) if tag == "rot" else None
    torch.set_num_threads(8)
    pred, loss, grads, stats = H.hm_train_step(left, right, gt, plen, sd, lam=1.0)
    np.testing.assert_allclose(pred.reshape(-1)[::97].numpy(), g["pred_sample"], atol=2e-5)
    names = ("limb_heatmap_left", "limb_heatmap_right") if tag == "rot" else ("heatmap_left", "heatmap_right")
    np.testing.assert_allclose(float(loss), float(g["loss_" + names[0]]) + float(g["loss_" + names[1]]), rtol=1e-5)
    norms = dict(zip(g["grad_keys"], g["grad_norms"]))
    assert sorted(k for k, v in grads.items() if v is not None) == sorted(g["grad_keys"])
    for k in g["grad_keys"]:
        gr = grads[k]
        got = gr.reshape(-1)[:: max(1, gr.numel() // 257)].numpy()
        scale = norms[k] / np.sqrt(gr.numel())
        assert np.abs(got - g["g:" + k]).max() <= 2e-2 * scale + 1e-9, k        # fp32 CPU vs fp32 CPU, different summation orders
        np.testing.assert_allclose(float(gr.double().norm()), norms[k], rtol=5e-3, err_msg=k)
    for k, v in stats.items():
        np.testing.assert_allclose(v.numpy(), g["buf:" + k], rtol=1e-4, atol=1e-6, err_msg=k)
