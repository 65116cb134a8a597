"""Deterministic synthetic weights and inputs (no torch RNG, no files).

Every tensor is a pure function of a string key and its shape: element ``i`` of
tensor ``key`` is ``splitmix64(fnv1a64(key) + i)`` mapped to a uniform in [0,1)
and then shaped by a per-kind rule (weights ~ U(-a, a) with variance
1/fan_in, norm gains near 1, running variances strictly positive ...).

The same function is used by the golden-vector generator (which runs next to
the reference in the build container), by the tests, by ``bench.py`` and by
``__graft_entry__.smoke()`` on the GPU box, so weights never have to be
stored: SURVEY.md section 8(c)/(d) "hash-RNG".
"""
from __future__ import annotations

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for ch in text.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(key: str, n: int) -> np.ndarray:
    """n float64 uniforms in [0,1) that depend only on (key, index)."""
    base = np.uint64(fnv1a64(key))
    out = np.empty(n, dtype=np.float64)
    step = 1 << 22
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        with np.errstate(over="ignore"):
            idx = np.arange(lo, hi, dtype=np.uint64) + base
        z = _splitmix64(idx)
        out[lo:hi] = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return out


def synth_tensor(key: str, shape, kind: str | None = None) -> np.ndarray:
    """Synthetic fp32 (or int64 for counters) value for a state_dict entry.

    kind is inferred from the key/shape when not given:
      counter      -> zeros int64            (num_batches_tracked)
      running_var  -> 0.5 + u                (strictly positive)
      running_mean -> 0.2 (u - 0.5)
      token        -> 0.2 (u - 0.5)          (cls/mask token, position embeddings)
      gain         -> 1 + 0.2 (u - 0.5)      (1-D ``weight`` of a norm layer)
      bias         -> 0.2 (u - 0.5)          (any 1-D ``bias``)
      weight       -> U(-a, a), a = sqrt(3 / fan_in)   (variance 1/fan_in)
    """
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    leaf = key.rsplit(".", 1)[-1]
    if kind is None:
        if leaf == "num_batches_tracked":
            kind = "counter"
        elif leaf == "running_var":
            kind = "running_var"
        elif leaf == "running_mean":
            kind = "running_mean"
        elif leaf in ("cls_token", "mask_token", "position_embeddings"):
            kind = "token"
        elif leaf == "bias":
            kind = "bias"
        elif len(shape) <= 1:
            kind = "gain"
        else:
            kind = "weight"
    if kind == "counter":
        return np.zeros(shape, dtype=np.int64)
    u = uniform01(key, n)
    if kind == "running_var":
        v = 0.5 + u
    elif kind == "running_mean":
        v = 0.2 * (u - 0.5)
    elif kind == "token":
        v = 0.2 * (u - 0.5)
    elif kind == "gain":
        v = 1.0 + 0.2 * (u - 0.5)
    elif kind == "bias":
        v = 0.2 * (u - 0.5)
    elif kind == "weight":
        fan_in = int(np.prod(shape[1:]))
        v = (u - 0.5) * (2.0 * np.sqrt(3.0 / fan_in))
    else:
        raise ValueError(f"unknown synthetic kind {kind!r}")
    return v.astype(np.float32).reshape(shape)


def synth_state_dict(spec, prefix: str = "", salt: str = ""):
    """spec: iterable of (key, shape). Returns {key: np.ndarray}.

    ``salt`` lets two networks with identical key names (pos / rot heatmap
    estimators) get different weights.
    """
    return {k: synth_tensor(salt + prefix + k, shp) for k, shp in spec}


def synth_input(name: str, shape, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    """Synthetic fp32 input: uniform in [lo, hi) keyed by ``name``."""
    n = int(np.prod(shape))
    u = uniform01("input:" + name, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def synth_hm_state_dict(n_hm_per_eye: int, salt: str, model_name: str = "resnet18"):
    """Hash-RNG state_dict of a heatmap estimator.  Backbone tensors appear under two keys in the reference's
    state_dict (spec.hm_state_spec); load_state_dict writes them in key order, so the value that survives is the one
    generated from the LATER (alias) key -- reproduced here so weights equal what tools/make_golden.py loaded."""
    from . import spec
    entries = spec.hm_state_spec(n_hm_per_eye, model_name)
    last_key_of = {}
    for key, shape, alias in entries:
        last_key_of[alias or key] = key
    out = {}
    for key, shape, alias in entries:
        canon = alias or key
        if canon not in out:
            out[canon] = synth_tensor(salt + last_key_of[canon], shape)
        out[key] = out[canon]
    return out
