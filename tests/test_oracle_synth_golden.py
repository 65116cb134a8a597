"""The heatmap-synthesis oracle (oracle/heatmap_synth_ref.py) against fixtures produced by the reference's own
coord2d_to_heatmap and get_limb_data (tools/make_golden.py gen_synth).  skimage's line_aa is not installed here: the fixture
was generated with the oracle's restatement of it plugged into the reference, so the line walk itself is unpinned."""
import os

import numpy as np
import pytest

from egotap_amd.synthetic import synth_input
from oracle import heatmap_synth_ref as R

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "heatmap_synth.npz"))


def inputs(preset, n):
    p2l = synth_input(f"synth_p2l_{preset}", (3, n, 2), -60.0, 1080.0).astype(np.float64)
    p2r = synth_input(f"synth_p2r_{preset}", (3, n, 2), -60.0, 1080.0).astype(np.float64)
    p2l[0, 1] = [512.0, 256.0]
    p2l[0, 2] = [-100.0, 500.0]
    p3 = synth_input(f"synth_p3_{preset}", (3, n, 3), -40.0, 40.0).astype(np.float64)
    return p2l, p2r, p3


@pytest.mark.parametrize("preset,n", [("UnrealEgo", 16), ("EgoCap", 18)])
@pytest.mark.parametrize("res", [64, 128])
def test_oracle_matches_reference_fixture(preset, n, res):
    p2l, p2r, p3 = inputs(preset, n)
    J = n - 1
    for b in range(3):
        cat, plen, theta = R.process_frame(p2l[b], p2r[b], p3[b], preset, res)
        tag = f"{preset}_{res}_{b}"
        np.testing.assert_array_equal(cat[:2 * J], GOLD[tag + "_pos"])
        limb = GOLD[tag + "_limb"] * np.float32(2)
        cos, sin = np.cos(GOLD[tag + "_theta"])[:, None, None], np.sin(GOLD[tag + "_theta"])[:, None, None]
        np.testing.assert_allclose(cat[2 * J:3 * J], limb[:J] * cos, atol=1e-7)
        np.testing.assert_allclose(cat[3 * J:4 * J], limb[:J] * sin, atol=1e-7)
        np.testing.assert_allclose(cat[4 * J:5 * J], limb[J:] * cos, atol=1e-7)
        np.testing.assert_allclose(plen, GOLD[tag + "_len"], rtol=1e-6)
        np.testing.assert_allclose(theta, GOLD[tag + "_theta"], rtol=1e-6)


def test_line_aa_properties():
    """end points get full weight, the walk is symmetric in length, a single pixel is a single sample"""
    rr, cc, val = R.line_aa(3, 4, 3, 4)
    assert list(rr) == [3] and list(cc) == [4] and val[0] == 1.0
    rr, cc, val = R.line_aa(0, 0, 10, 0)
    assert set(zip(rr, cc)) >= {(i, 0) for i in range(11)} and np.all(val[(cc == 0)] == 1.0)
    rr, cc, val = R.line_aa(2, 3, 40, 17)
    assert (rr[0], cc[0]) == (2, 3) and (40, 17) in set(zip(rr, cc)) and np.all((val >= 0) & (val <= 1))


def test_line_aa_published_example():
    """The one known answer available without skimage: the example in skimage.draw.line_aa's published docstring
    (line_aa(1, 1, 8, 8) painted as val * 255 into a 10x10 uint8 image: 255 on the diagonal, 74 beside it).  A single
    diagonal vector -- the line walk stays 'parity unpinned' for general slopes (oracle header)."""
    rr, cc, val = R.line_aa(1, 1, 8, 8)
    img = np.zeros((10, 10), np.uint8)
    img[rr, cc] = val * 255
    want = np.zeros((10, 10), np.uint8)
    for i in range(1, 9):
        want[i, i] = 255
        if i < 8:
            want[i, i + 1] = want[i + 1, i] = 74
    np.testing.assert_array_equal(img, want)
