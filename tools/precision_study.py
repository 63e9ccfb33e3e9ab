#!/usr/bin/env python3
"""CPU emulation of per-layer operand precisions for the unet_2 forward (no GPU, no product code path).

Question it answers: which of the f16f8 mode's residual cross terms are needed to stay under the 1e-4 MAE gate on the
'he' formula weights?  Per MFMA layer the operands can be
    'x'   exact fp32 (stands for f16 + fp8 residual; the residual error, ~2^-15 relative, is emulated too with 'r')
    'h'   rounded to f16 (round to nearest even), products exact, fp32 accumulation
for the weights and, separately, the activations.  Self-contained (torch.nn.functional only).

    python tools/precision_study.py [batch] [size]
"""
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from ws_unet_amd import formula  # noqa: E402

LAYERS = ["e12", "e21", "e22", "e31", "e32", "upconv3", "d31", "d32", "upconv4", "d41", "d42"]


def q16(t):
    return t.to(torch.float16).to(torch.float32)


def q8res(t):
    """f16 + e4m3 residual at a fixed 2^12 scale, as the f16f8 storage does."""
    h = q16(t)
    r = ((t - h) * 4096.0).to(torch.float8_e4m3fn).to(torch.float32) / 4096.0
    return h + r


def prep(t, how):
    if how == "x":
        return t
    if how == "h":
        return q16(t)
    if how == "r":
        return q8res(t)
    raise ValueError(how)


def forward(x, sd, cfg):
    """cfg[layer] = (weight precision, activation precision)."""
    def cv(n, v):
        wq, xq = cfg.get(n, ("x", "x"))
        v = prep(v, xq)
        w = prep(sd[n + ".weight"], wq)
        return F.relu(F.conv2d(F.pad(v, (1, 1, 1, 1), mode="reflect"), w, sd[n + ".bias"]))

    def up(n, v):
        wq, xq = cfg.get(n, ("x", "x"))
        return F.conv_transpose2d(prep(v, xq), prep(sd[n + ".weight"], wq), sd[n + ".bias"], stride=2)

    xe11 = cv("e11", x)
    xe12 = cv("e12", xe11)
    xe21 = cv("e21", F.max_pool2d(xe12, 2, 2))
    xe22 = cv("e22", xe21)
    xe31 = cv("e31", F.max_pool2d(xe22, 2, 2))
    xe32 = cv("e32", xe31)
    xd31 = cv("d31", torch.cat([up("upconv3", xe32), xe22], 1))
    xd32 = cv("d32", xd31)
    xd41 = cv("d41", torch.cat([up("upconv4", xd32), xe12], 1))
    xd42 = cv("d42", xd41)
    return torch.sigmoid(F.conv2d(xd42, sd["outconv.weight"], sd["outconv.bias"]))


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    torch.set_num_threads(8)
    sd = {k: torch.from_numpy(v) for k, v in formula.formula_state_dict(2, "he").items()}
    u8 = formula.synthetic_images(batch, size, size, seed=1000)
    x = torch.from_numpy(u8.astype(np.float32) / np.float32(255.))[:, None]
    with torch.no_grad():
        t0 = time.time()
        ref = forward(x, sd, {})
        print(f"reference forward {time.time() - t0:.1f} s; out mean {ref.mean():.4f} std {ref.std():.4f}", flush=True)

        def mae(cfg):
            return (forward(x, sd, cfg) - ref).abs().mean().item()

        print("all layers (w,x):")
        for wq, xq in [("h", "h"), ("x", "h"), ("h", "x"), ("r", "r"), ("r", "h"), ("x", "x")]:
            print(f"  w={wq} x={xq}: MAE {mae({n: (wq, xq) for n in LAYERS}):.3e}", flush=True)
        print("single layer in plain f16 (others exact):  w-only / x-only / both")
        for n in LAYERS:
            a = mae({n: ("h", "x")}); b = mae({n: ("x", "h")}); c = mae({n: ("h", "h")})
            print(f"  {n:8s} {a:.3e} {b:.3e} {c:.3e}", flush=True)


if __name__ == "__main__":
    main()
