"""Deterministic, fully specified generators for weights, images and stego pairs.

The reference ships no pretrained UNet checkpoints, so parity is
pinned on *formula* weights: every tensor element is a pure function of
(tensor name, flat index, variant), computed with 64-bit integer hashing
(splitmix64 finaliser) -- no RNG library state, identical on every machine and
numpy version.  Goldens under tests/golden/ were produced by loading these
weights into the reference model (tests/golden/make_golden.py).

Variants
--------
``default``  bound = 1/sqrt(fan_in) for weight and bias: the distribution of
             PyTorch's default Conv2d init that the reference relies on
             (src/unet/model/unet.py:82-135 construct plain nn.Conv2d).  With
             it the network output is almost constant (0.504 +- 2e-4), so it is
             a weak parity test.
``he``       weight bound = sqrt(6/fan_in) (variance preserving through ReLU),
             bias bound 0.1.  Output spans (0,1) like a trained predictor; this
             is the variant the 1e-4 MAE gate is evaluated on.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hash_u64(key: int, n: int, offset: int = 0) -> np.ndarray:
    """n 64-bit hashes of indices offset..offset+n-1 under ``key``."""
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = idx * _GOLD + np.uint64(key & 0xFFFFFFFFFFFFFFFF)
    return _mix64(_mix64(z) + _GOLD)


def uniform_pm1(key: int, n: int) -> np.ndarray:
    """float64 uniform in (-1, 1) on a 24-bit lattice."""
    u24 = (hash_u64(key, n) >> np.uint64(40)).astype(np.float64)
    return (u24 + 0.5) / 8388608.0 - 1.0


def formula_tensor(name: str, shape: Tuple[int, ...], bound: float, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    key = fnv1a64(f"{name}#{seed}")
    return (uniform_pm1(key, n) * float(bound)).astype(np.float32).reshape(shape)


# ---------------------------------------------------------------------------
# UNet state-dict shapes: reference src/unet/model/unet.py:82-135
# ---------------------------------------------------------------------------

def unet_param_shapes(nsteps: int, in_channels: int = 1, out_channels: int = 1) -> "OrderedDict[str, Tuple[int, ...]]":
    """Key order == the reference module's ``state_dict()`` order (registration
    order in unet.py:82-135: encoder levels, then upconv1/d11/d12 ... outconv)."""
    assert 0 <= nsteps <= 4
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(name, cin, cout, k=3):
        sh[name + ".weight"] = (cout, cin, k, k)
        sh[name + ".bias"] = (cout,)

    def convt(name, cin, cout):
        sh[name + ".weight"] = (cin, cout, 2, 2)
        sh[name + ".bias"] = (cout,)

    conv("e11", in_channels, 64); conv("e12", 64, 64)
    if nsteps >= 1:
        conv("e21", 64, 128); conv("e22", 128, 128)
    if nsteps >= 2:
        conv("e31", 128, 256); conv("e32", 256, 256)
    if nsteps >= 3:
        conv("e41", 256, 512); conv("e42", 512, 512)
    if nsteps >= 4:
        conv("e51", 512, 1024); conv("e52", 1024, 1024)
    if nsteps >= 4:
        convt("upconv1", 1024, 512); conv("d11", 1024, 512); conv("d12", 512, 512)
    if nsteps >= 3:
        convt("upconv2", 512, 256); conv("d21", 512, 256); conv("d22", 256, 256)
    if nsteps >= 2:
        convt("upconv3", 256, 128); conv("d31", 256, 128); conv("d32", 128, 128)
    if nsteps >= 1:
        convt("upconv4", 128, 64); conv("d41", 128, 64); conv("d42", 64, 64)
    conv("outconv", 64, out_channels, k=1)
    return sh


def formula_state_dict(nsteps: int, variant: str = "he", seed: int = 0,
                       in_channels: int = 1, out_channels: int = 1) -> "OrderedDict[str, np.ndarray]":
    """Formula weights as numpy float32 arrays keyed like the reference state_dict."""
    if variant not in ("default", "he"):
        raise ValueError(variant)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    shapes = unet_param_shapes(nsteps, in_channels, out_channels)
    for name, shape in shapes.items():
        layer, kind = name.rsplit(".", 1)
        wshape = shapes[layer + ".weight"]
        if layer.startswith("upconv"):
            fan_in = wshape[0]                      # Cin terms per output element
        else:
            fan_in = wshape[1] * wshape[2] * wshape[3]
        if variant == "default":
            bound = 1.0 / math.sqrt(fan_in)
        elif kind == "weight":
            bound = math.sqrt(6.0 / fan_in)
        else:
            bound = 0.1
        out[name] = formula_tensor(f"{variant}/unet_{nsteps}/{name}", shape, bound, seed)
    return out


# ---------------------------------------------------------------------------
# Synthetic images / stego pairs
# ---------------------------------------------------------------------------

def synthetic_images(n: int, h: int, w: int, seed: int = 12345, smooth: bool = True) -> np.ndarray:
    """uint8 (n,h,w).  ``smooth``: 5x5 reflect box blur of hashed uniform noise
    in exact integer arithmetic (round-half-up), else raw hashed noise."""
    key = fnv1a64(f"img#{seed}#{h}x{w}")
    raw = (hash_u64(key, n * h * w) >> np.uint64(56)).astype(np.int64).reshape(n, h, w)
    if not smooth:
        return raw.astype(np.uint8)
    p = np.pad(raw, ((0, 0), (2, 2), (2, 2)), mode="reflect")
    acc = np.zeros((n, h, w), dtype=np.int64)
    for dy in range(5):
        for dx in range(5):
            acc += p[:, dy:dy + h, dx:dx + w]
    # stretch contrast x3 around 128 so the image keeps texture, then clip
    v = (acc * 2 + 25) // 50                      # rounded mean
    v = np.clip((v - 128) * 3 + 128, 0, 255)
    return v.astype(np.uint8)


def lsbr_embed(cover: np.ndarray, alpha: float, seed: int = 777) -> np.ndarray:
    """LSB-replacement simulator with change rate beta = alpha/2: flips the LSB
    of each pixel independently with probability alpha/2 (data invariant checked
    in SURVEY.md section 4: reference stego differs from cover only in LSBs, rate alpha/2)."""
    assert cover.dtype == np.uint8
    key = fnv1a64(f"lsbr#{seed}#{cover.shape}")
    u = (hash_u64(key, cover.size) >> np.uint64(32)).astype(np.float64) / 4294967296.0
    flip = (u < alpha / 2.0).reshape(cover.shape)
    return (cover ^ flip.astype(np.uint8)).astype(np.uint8)


def bernoulli_mask(shape: Tuple[int, ...], keep_prob: float, seed: int = 4242) -> np.ndarray:
    """float32 keep-mask (1 = keep) used to drive UniformDropout deterministically."""
    key = fnv1a64(f"mask#{seed}#{shape}")
    n = int(np.prod(shape))
    u = (hash_u64(key, n) >> np.uint64(32)).astype(np.float64) / 4294967296.0
    return (u < keep_prob).astype(np.float32).reshape(shape)
