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
