"""CPU restatement (numpy + scipy) of the reference's ground-truth heatmap synthesis -- TEST INFRASTRUCTURE ONLY: imported by
tests/ and tools/make_golden.py, never by the product path.

  coord2d_to_heatmap     utils/projection.py:263-279
  get_limb_data ('line') utils/data.py:175-252  (+ get_line_limb_heatmap :176-186)
  process_frame          dataloader/data_loader.py:76-215 (the heatmap / limb / theta / pixel-length parts, heatmap_type 'sin')

Pinned: tests/golden/heatmap_synth.npz is produced by the reference's own coord2d_to_heatmap and get_limb_data
(tools/make_golden.py gen_synth).  skimage is NOT installed here, so the reference's `from skimage.draw import line_aa` is
satisfied with ``line_aa`` below, a restatement of skimage's published _line_aa (skimage/draw/_draw.pyx, Zingl's anti-aliased
Bresenham): **parity unpinned for that one step** (its only check is the single example of skimage's published docstring,
tests/test_oracle_synth_golden.py::test_line_aa_published_example); the Gaussian filter is scipy's own (the reference's dependency).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.ndimage import gaussian_filter

KINEMATIC_PARENTS = {          # utils/util.py:51-52
    "UnrealEgo": [0, 0, 1, 1, 2, 3, 4, 5, 2, 3, 8, 9, 10, 11, 12, 13],
    "EgoCap": [0, 0, 1, 2, 3, 4, 1, 6, 7, 8, 2, 10, 11, 12, 6, 14, 15, 16],
}


def line_aa(r0, c0, r1, c1):
    """skimage.draw.line_aa: (rr, cc, val); C float arithmetic for err / ed as in the Cython source"""
    f = np.float32
    rr, cc, val = [], [], []
    dc, dr = abs(c0 - c1), abs(r0 - r1)
    err = f(dc - dr)
    sign_c = 1 if c0 < c1 else -1
    sign_r = 1 if r0 < r1 else -1
    ed = f(1.0) if dc + dr == 0 else f(math.sqrt(dc * dc + dr * dr))
    c, r = c0, r0
    while True:
        cc.append(c); rr.append(r); val.append(f(abs(f(err - f(dc)) + f(dr))) / ed)
        err_prime, c_prime = err, c
        if f(2.0) * err_prime >= f(-dc):
            if c == c1:
                break
            if err_prime + f(dr) < ed:
                cc.append(c); rr.append(r + sign_r); val.append(f(abs(err_prime + f(dr))) / ed)
            err = f(err - f(dr))
            c += sign_c
        if f(2.0) * err_prime <= f(dr):
            if r == r1:
                break
            if f(dc) - err_prime < ed:
                cc.append(c_prime + sign_c); rr.append(r); val.append(f(abs(f(dc) - err_prime)) / ed)
            err = f(err + f(dc))
            r += sign_r
    return np.array(rr, dtype=np.intp), np.array(cc, dtype=np.intp), 1.0 - np.array(val, dtype=float)


def coord2d_to_heatmap(coord2d, res=64, sigma=1.0):
    hm = np.zeros((coord2d.shape[0], res, res), dtype=np.float32)
    margin = int(4 * sigma)
    for i in range(coord2d.shape[0]):
        pos = coord2d[i] / 1024.0 * res
        x, y = pos[0], pos[1]
        big = np.zeros((res + 2 * margin, res + 2 * margin), dtype=np.float32)
        if -4 <= y < res + 4 and -4 <= x < res:
            big[int(y) + margin, int(x) + margin] = 1.0
        big = gaussian_filter(big, sigma=sigma)
        hm[i] = big[margin:-margin, margin:-margin]
    hm /= 0.15915589174187972
    return hm


def get_limb_data(pts2d, pts3d, res=64, sigma=1, joint_preset="UnrealEgo"):
    parents = KINEMATIC_PARENTS[joint_preset]
    n = len(parents)
    maps = np.zeros((n - 1, res, res), dtype=np.float32)
    lengths = np.zeros(n - 1, dtype=np.float32)
    theta = np.zeros(n - 1, dtype=np.float32)
    for j in range(1, n):
        par = parents[j]
        div = 1024.0 / res
        p, c = pts2d[par] / div, pts2d[j] / div
        limb = pts3d[par] - pts3d[j]
        theta[j - 1] = np.arctan(limb[2] / np.linalg.norm(limb[:2]))
        lengths[j - 1] = np.linalg.norm(p - c) + 1.0
        hm = np.zeros((res, res), dtype=np.float32)
        pi, ci = np.rint(p).astype(int), np.rint(c).astype(int)
        rr, cc, val = line_aa(pi[0], pi[1], ci[0], ci[1])
        ok = (rr >= 0) & (rr <= res - 1) & (cc >= 0) & (cc <= res - 1)
        hm[cc[ok], rr[ok]] = val[ok]
        hm = gaussian_filter(hm, sigma=sigma, mode="constant")
        hm *= sigma
        maps[j - 1] = hm
    return maps, lengths, theta


def process_frame(pts2d_left, pts2d_right, local_pose, joint_preset="UnrealEgo", res=64):
    """one frame -> (cat [6J, res, res], plength [2, J], theta [J]); channel order L pos, R pos, L cos, L sin, R cos, R sin"""
    hl = coord2d_to_heatmap(pts2d_left[1:], res)
    hr = coord2d_to_heatmap(pts2d_right[1:], res)
    ll, len_l, theta = get_limb_data(pts2d_left, local_pose, res, 1, joint_preset)
    lr, len_r, _ = get_limb_data(pts2d_right, local_pose, res, 1, joint_preset)
    ll, lr = ll * np.float32(2), lr * np.float32(2)
    cos, sin = np.cos(theta)[:, None, None].astype(np.float32), np.sin(theta)[:, None, None].astype(np.float32)
    cat = np.concatenate([hl, hr, ll * cos, ll * sin, lr * cos, lr * sin], axis=0).astype(np.float32)
    return cat, np.stack([len_l, len_r]), theta
