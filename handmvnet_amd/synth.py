"""Deterministic, portable synthetic weights and inputs.

No trained or pretrained weights are obtainable offline, so
every parity and benchmark run uses weights generated here.  The generator is a pure
integer counter hash (splitmix64) followed by exact float arithmetic (adds/multiplies
only, no libm), so the same bytes come out in this container, on the GPU box and in
any other numpy -- nothing 86 MB large has to be shipped.

Distributions follow SURVEY.md section 4 item 4: BN statistics/affine are randomised so
a BN-folding bug cannot hide behind identity BN, biases are non-zero, conv/linear
weights are He/Xavier-scaled so activations stay O(1) through 16 residual blocks.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Tuple

import numpy as np

from .spec import HotPathConfig, state_dict_layout

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + _GOLD).astype(np.uint64)
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        return x ^ (x >> np.uint64(31))


def _raw(key: str, seed: int, n: int, lane: int = 0) -> np.ndarray:
    base = np.uint64((zlib.crc32(key.encode()) << 20) ^ (seed * 0x51ED270B + lane * 0x2545F491) & 0xFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * _GOLD + _splitmix(np.array([base], dtype=np.uint64))[0]
    return _splitmix(ctr)


def uniform01(key: str, seed: int, n: int, lane: int = 0) -> np.ndarray:
    """n floats in [0,1) with 24 random bits each (exact in fp32)."""
    return ((_raw(key, seed, n, lane) >> np.uint64(40)).astype(np.float64)) * (1.0 / (1 << 24))


def normalish(key: str, seed: int, n: int) -> np.ndarray:
    """Irwin-Hall(4) centred and scaled to unit variance: only adds/multiplies."""
    s = uniform01(key, seed, n, 1) + uniform01(key, seed, n, 2) + uniform01(key, seed, n, 3) + uniform01(key, seed, n, 4)
    return (s - 2.0) * 1.7320508075688772


def _tensor(key: str, shape: tuple, seed: int, cfg: HotPathConfig) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros((), dtype=np.int64)
    if leaf == "running_mean":
        v = 0.1 * normalish(key, seed, n)
    elif leaf == "running_var":
        v = 0.5 + uniform01(key, seed, n)
    elif len(shape) == 1 and leaf == "weight":          # BN gamma / LayerNorm weight
        v = 0.5 + uniform01(key, seed, n)
        # last BN of a residual branch: keep the residual stream from growing block by block
        if cfg.is_hrnet:
            if (".layer1." in key and ".bn3." in key) or (".branches." in key and ".bn2." in key):
                v = v * 0.25
            elif ".fuse_layers." in key:
                v = v * 0.5   # up to 4 fuse terms are summed per branch
        elif key.startswith("backbone.layer") and (
                (cfg.is_paper and ".bn3." in key) or (not cfg.is_paper and ".bn2." in key)):
            v = v * 0.25
    elif leaf == "bias":
        v = 0.1 * normalish(key, seed, n)
    elif len(shape) == 4 and key.startswith("joints_decoder"):   # ChebConv [K+1, 1, in, out]
        v = normalish(key, seed, n) * np.sqrt(2.0 / (shape[2] + shape[3]))
    elif len(shape) == 4:
        if key == "pose_net.0.weight" and not cfg.is_paper:      # ConvTranspose2d [in, out, 4, 4], stride 2
            fan_in = shape[0] * shape[2] * shape[3] / 4.0
        else:
            fan_in = shape[1] * shape[2] * shape[3]
        gain = 2.0
        last_hm = "pose_net.weight" if cfg.is_hrnet else ("pose_net.3.weight" if cfg.is_paper else "pose_net.6.weight")
        if key == last_hm:
            # heat-map std ~0.03 -> x1000 temperature gives logits of std ~30: mostly one-hot
            # joints with a healthy tail of genuinely soft (sub-pixel) ones
            gain = 2.0e-4
        elif key.startswith("sample_nets"):
            gain = 0.5
        v = normalish(key, seed, n) * np.sqrt(gain / fan_in)
    elif len(shape) == 2:
        v = normalish(key, seed, n) * np.sqrt(1.0 / shape[1])
    elif leaf == "probe":                                # nn.Parameter(torch.randn(1, 21, d)), layers.py:250
        v = normalish(key, seed, n)
    else:
        raise ValueError(f"no rule for {key} {shape}")
    return v.astype(np.float32).reshape(shape)


def synth_state_dict(cfg: HotPathConfig, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    return OrderedDict((k, _tensor(k, s, seed, cfg)) for k, s in state_dict_layout(cfg).items())


def synth_inputs(cfg: HotPathConfig, batch: int, seed: int = 0, size: int | None = None
                 ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """x [B,V,3,S,S] ~ N(0,1); bbox [B,V,4] valid (x1,y1,x2,y2) px; intrinsic [B,V,4] (fx,fy,cx,cy).
    Deliberately NOT eval_fps.py's randn bbox / uninitialised intrinsics."""
    s = size or cfg.image_size
    v = cfg.num_views
    x = normalish("input.x", seed, batch * v * 3 * s * s).astype(np.float32).reshape(batch, v, 3, s, s)
    u = uniform01("input.bbox", seed, batch * v * 3).reshape(batch, v, 3)
    x1, y1, side = 200.0 * u[..., 0], 200.0 * u[..., 1], 80.0 + 220.0 * u[..., 2]
    bbox = np.stack([x1, y1, x1 + side, y1 + side], axis=-1).astype(np.float32)
    w = uniform01("input.intr", seed, batch * v * 4).reshape(batch, v, 4)
    intr = np.stack([400.0 + 500.0 * w[..., 0], 400.0 + 500.0 * w[..., 1],
                     300.0 + 40.0 * w[..., 2], 220.0 + 40.0 * w[..., 3]], axis=-1).astype(np.float32)
    return x, bbox, intr
