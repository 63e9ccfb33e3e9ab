#!/usr/bin/env python3
"""CPU emulation (no GPU, no product code): what would BLOCK-SCALED fp6 / fp4 cross terms cost in accuracy?

The f16f8 arithmetic multiplies  w x ~ f16(w) f16(x) + Q(w - f16 w) Q(x) + Q(w) Q(x - f16 x)  with Q = e4m3 at fixed power-of-two scales (the
operands' 18 binades of range make a fixed scale enough).  The fp6 / fp4 forms of v_mfma_scale_f32_32x32x64_f8f6f4 run at twice the fp8 rate
(profiles/r03/conv3x3_pl_units_probe.md: -6.5 .. -16 % kernel time) but hold 4-7 binades: they need the instruction's per-block E8M0 scales --
here per (pixel, 16-channel chunk) for the activations and per (output channel, tap, 16-channel chunk) for the weights, shared by the copy and
the (pre-scaled) residual half of a block, which is what a K-block of 32 = one tap x 16 channels x both terms would carry.

    python tools/precision_study_lowbit.py [batch] [size]
prints the MAE of the unet_2 output ('he' formula weights) against exact fp32 for Q in {e4m3 fixed (today), e2m3, e3m2, e2m1 block-scaled}."""
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from ws_unet_amd import formula  # noqa: E402

FORMATS = {"e2m3": (2, 3, 7.5), "e3m2": (3, 2, 28.0), "e2m1": (2, 1, 6.0)}      # exponent bits, mantissa bits, largest value


def q16(t):
    return t.to(torch.float16).to(torch.float32)


def q_small(v, fmt):
    """round v (already divided by its block scale) to the format's grid: normals 2^e (1 + m / 2^mbits), subnormals below 2^emin, saturating"""
    eb, mb, vmax = FORMATS[fmt]
    bias = (1 << (eb - 1)) - 1
    emin = 1 - bias
    a = v.abs().clamp_min(1e-38)
    e = torch.floor(torch.log2(a)).clamp_min(emin)
    step = torch.exp2(e - mb)
    q = torch.round(v.abs() / step) * step
    return torch.sign(v) * q.clamp_max(vmax)


HW_RULE = False          # True: the scale rule a loader can afford -- 2^(floor(log2 amax) - 1) from the exponent field of the block's largest f16 (amax / scale in [2, 4))


def block_scale(amax, fmt):
    """E8M0 scale 2^E with amax / 2^E <= the format's largest value (no saturation), as large a use of the range as a power of two allows"""
    if HW_RULE:
        e = torch.floor(torch.log2(amax.clamp_min(2.0 ** -14)))
        if HW_RULE == 2:     # one compare more: a block whose largest f16 has a mantissa <= 1.5 takes the next smaller scale (amax / scale in [4, 6] instead of [2, 3])
            return torch.exp2(e - 1 - (amax <= 1.5 * torch.exp2(e)).to(amax.dtype))
        return torch.exp2(e - 1)
    vmax = FORMATS[fmt][2]
    return torch.exp2(torch.ceil(torch.log2(amax.clamp_min(1e-30) / vmax)))


def quant_pair(c, r, fmt, block_dims):
    """copy c and pre-scaled residual r share one scale per block (block = all of `block_dims` of a 16-channel chunk view)"""
    amax = c.abs().amax(dim=block_dims, keepdim=True) if HW_RULE else torch.maximum(c.abs().amax(dim=block_dims, keepdim=True), r.abs().amax(dim=block_dims, keepdim=True))
    s = block_scale(amax, fmt)
    return q_small(c / s, fmt) * s, q_small(r / s, fmt) * s


def split_x(x, fmt):
    """activations (N, C, H, W): returns f16 part, Q(copy), Q(residual) (true scale)"""
    h = q16(x)
    r = x - h
    if HW_RULE and fmt != "e4m3":
        r = (r * 4096).to(torch.float8_e4m3fn).to(torch.float32) / 4096          # the loader sees the STORED e4m3 residual
    if fmt == "e4m3":
        c8 = (h / 4).to(torch.float8_e4m3fn).to(torch.float32) * 4
        r8 = (r * 4096).to(torch.float8_e4m3fn).to(torch.float32) / 4096
        return h, c8, r8
    n, c, hh, ww = x.shape
    pad = (-c) % 16
    hv = F.pad(h, (0, 0, 0, 0, 0, pad)).view(n, -1, 16, hh, ww)
    rv = F.pad(r, (0, 0, 0, 0, 0, pad)).view(n, -1, 16, hh, ww) * 2048.0     # |r| <= 2^-11 |x|: the halves of a block in one range
    cq, rq = quant_pair(hv, rv, fmt, (2,))
    return h, cq.view(n, -1, hh, ww)[:, :c], (rq / 2048.0).view(n, -1, hh, ww)[:, :c]


def split_w(w, fmt):
    """weights (Cout, Cin, kh, kw): blocks = (cout, tap, 16 input channels)"""
    h = q16(w)
    r = w - h
    if fmt == "e4m3":
        c8 = (h * 64).to(torch.float8_e4m3fn).to(torch.float32) / 64
        r8 = (r * 262144).to(torch.float8_e4m3fn).to(torch.float32) / 262144
        return h, c8, r8
    co, ci, kh, kw = w.shape
    pad = (-ci) % 16
    hv = F.pad(h, (0, 0, 0, 0, 0, pad)).view(co, -1, 16, kh, kw)
    rv = F.pad(r, (0, 0, 0, 0, 0, pad)).view(co, -1, 16, kh, kw) * 2048.0
    cq, rq = quant_pair(hv, rv, fmt, (2,))
    return h, cq.view(co, -1, kh, kw)[:, :ci], (rq / 2048.0).view(co, -1, kh, kw)[:, :ci]


LAYER_FMT = {}           # per-layer override of the cross-term format (sensitivity study: `--per-layer`)


def forward(x, sd, fmt):
    def stored(t):                      # what the planar format keeps of an activation: f16 + e4m3 residual
        if fmt is None:
            return t
        h = q16(t)
        return h + ((t - h) * 4096).to(torch.float8_e4m3fn).to(torch.float32) / 4096

    def cv(n, v):
        w, b = sd[n + ".weight"], sd[n + ".bias"]
        vp = F.pad(v, (1, 1, 1, 1), mode="reflect")
        if fmt is None or n == "e11":
            y = F.conv2d(vp, w, b)
        else:
            f = LAYER_FMT.get(n, fmt)
            xh, xc, xr = split_x(vp, f)
            wh, wc, wr = split_w(w, f)
            y = F.conv2d(xh, wh, b) + F.conv2d(xc, wr) + F.conv2d(xr, wc)
        return stored(F.relu(y))

    def up(n, v):                       # the transposed convs keep today's e4m3 cross terms (their kernel is HBM-bound)
        w, b = sd[n + ".weight"], sd[n + ".bias"]
        if fmt is None:
            return F.conv_transpose2d(v, w, b, stride=2)
        xh, xc, xr = split_x(v, "e4m3")
        wt = w.permute(1, 0, 2, 3)
        wh, wc, wr = (t.permute(1, 0, 2, 3) for t in split_w(wt, "e4m3"))
        return stored(F.conv_transpose2d(xh, wh, b, stride=2) + F.conv_transpose2d(xc, wr, stride=2) + F.conv_transpose2d(xr, wc, stride=2))

    xe12 = cv("e12", cv("e11", x))
    xe22 = cv("e22", cv("e21", F.max_pool2d(xe12, 2, 2)))
    xe32 = cv("e32", cv("e31", F.max_pool2d(xe22, 2, 2)))
    xd32 = cv("d32", cv("d31", torch.cat([up("upconv3", xe32), xe22], 1)))
    w, b = sd["d42.weight"], sd["d42.bias"]
    xd41 = cv("d41", torch.cat([up("upconv4", xd32), xe12], 1))
    vp = F.pad(xd41, (1, 1, 1, 1), mode="reflect")
    if fmt is None:
        y = F.conv2d(vp, w, b)
    else:
        f = LAYER_FMT.get("d42", fmt)
        xh, xc, xr = split_x(vp, f)
        wh, wc, wr = split_w(w, f)
        y = F.conv2d(xh, wh, b) + F.conv2d(xc, wr) + F.conv2d(xr, wc)
    return torch.sigmoid(F.conv2d(F.relu(y), sd["outconv.weight"], sd["outconv.bias"]))     # the head reads the fp32 accumulators


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    batch = int(pos[0]) if len(pos) > 0 else 2
    size = int(pos[1]) if len(pos) > 1 else 256
    torch.set_num_threads(8)
    sd = {k: torch.from_numpy(v) for k, v in formula.formula_state_dict(2, "he").items()}
    u8 = formula.synthetic_images(batch, size, size, seed=1000)
    x = torch.from_numpy(u8.astype(np.float32) / np.float32(255.))[:, None]
    with torch.no_grad():
        ref = forward(x.double(), {k: v.double() for k, v in sd.items()}, None).float()
        print(f"unet_2 'he' weights, {batch} x {size} x {size}: output mean {ref.mean():.4f} std {ref.std():.4f}")
        global HW_RULE
        for fmt, hw in (("e4m3", False), ("e2m3", False), ("e3m2", False), ("e2m1", False), ("e2m1", True), ("e2m1", 2), ("e2m3", True)):
            t0 = time.time()
            HW_RULE = hw
            y = forward(x, sd, fmt)
            d = (y - ref).abs()
            what = " (fixed scales, today)" if fmt == "e4m3" else " (block scales, exponent-field rule + mantissa test, stored residuals)" if hw == 2 else " (block scales, exponent-field rule, stored residuals)" if hw else " (block scales)"
            print(f"  cross terms in {fmt:5s}{what}: MAE {d.mean().item():.3e}  max {d.max().item():.3e}   [{time.time() - t0:.0f} s]", flush=True)
        if "--per-layer" in sys.argv:       # the product's fp4 arithmetic with ONE layer (or only one layer) kept in e4m3: where the 2.1e-5 comes from
            HW_RULE = 2
            layers = ["e12", "e21", "e22", "e31", "e32", "d31", "d32", "d41", "d42"]
            for only in (False, True):
                for L in layers:
                    LAYER_FMT.clear()
                    LAYER_FMT.update({k: "e4m3" for k in layers if k != L} if only else {L: "e4m3"})
                    d = (forward(x, sd, "e2m1") - ref).abs()
                    print(f"  fp4 {'ONLY in' if only else 'everywhere but'} {L}: MAE {d.mean().item():.3e}", flush=True)
            LAYER_FMT.clear()


if __name__ == "__main__":
    main()
