"""Plain-numpy restatement of every operator on the path (TEST INFRASTRUCTURE ONLY).

Independent of torch: used to cross-check oracle/unet_ref.py on small cases and
to pin the integer / index semantics (reflect indices, pool tie order, LSB flip).
All tensors are NCHW float32 like the reference.  Accumulation is float64 unless
stated, results are cast to float32 at the end (so it is a *tight* reference, not
a bit-exact model of ATen's summation order).

Reference lines:
  reflect conv      src/unet/model/unet.py:73,82-132 (nn.Conv2d padding_mode='reflect')
  max-pool          unet.py:86,93
  transposed conv   unet.py:74,125,130
  UniformDropout    unet.py:32-42
  WS statistic      src/unet/evaluate.py:125-132
  L1 / WS / L1WS    src/_defs/losses.py:28-36,45-90,99-115
  AdamW             torch.optim.AdamW as used at src/detector/train.py:228
"""
from __future__ import annotations

import numpy as np


def reflect_index(i: np.ndarray, n: int) -> np.ndarray:
    """PyTorch 'reflect' (no edge repeat): -1 -> 1, n -> n-2."""
    i = np.where(i < 0, -i, i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def reflect_pad1(x: np.ndarray) -> np.ndarray:
    n, c, h, w = x.shape
    assert h >= 2 and w >= 2, "reflect pad 1 needs dims >= 2"
    yi = reflect_index(np.arange(-1, h + 1), h)
    xi = reflect_index(np.arange(-1, w + 1), w)
    return x[:, :, yi][:, :, :, xi]


def conv3x3_reflect(x, w, b=None, relu=False):
    """y[n,co,i,j] = b[co] + sum_{ci,u,v} w[co,ci,u,v] * xpad[n,ci,i+u,j+v]  (cross-correlation)."""
    n, c, h, wd = x.shape
    xp = reflect_pad1(x).astype(np.float64)
    w64 = w.astype(np.float64)
    y = np.zeros((n, w.shape[0], h, wd), dtype=np.float64)
    for u in range(3):
        for v in range(3):
            y += np.einsum("nchw,oc->nohw", xp[:, :, u:u + h, v:v + wd], w64[:, :, u, v], optimize=True)
    if b is not None:
        y += b.astype(np.float64)[None, :, None, None]
    if relu:
        y = np.maximum(y, 0)
    return y.astype(np.float32)


def conv1x1(x, w, b):
    y = np.einsum("nchw,oc->nohw", x.astype(np.float64), w[:, :, 0, 0].astype(np.float64))
    return (y + b.astype(np.float64)[None, :, None, None]).astype(np.float32)


def sigmoid(z):
    z = z.astype(np.float64)
    return (1.0 / (1.0 + np.exp(-z))).astype(np.float32)


def maxpool2x2(x):
    """Returns (pooled, argmax) with argmax in {0,1,2,3} = window position in
    row-major order (0,0),(0,1),(1,0),(1,1); ties -> FIRST max."""
    n, c, h, w = x.shape
    win = np.stack([x[:, :, 0::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 0::2], x[:, :, 1::2, 1::2]], axis=-1)
    arg = np.argmax(win, axis=-1)                 # numpy argmax returns the first maximum
    return np.max(win, axis=-1), arg.astype(np.uint8)


def maxpool2x2_backward(dy, arg, in_shape):
    dx = np.zeros(in_shape, dtype=dy.dtype)
    for k, (a, b) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
        dx[:, :, a::2, b::2] = np.where(arg == k, dy, 0)
    return dx


def convT2x2s2(x, w, b):
    """y[n,co,2i+a,2j+b] = bias[co] + sum_ci x[n,ci,i,j] * w[ci,co,a,b]; w is (Cin,Cout,2,2)."""
    n, c, h, wd = x.shape
    cout = w.shape[1]
    y = np.zeros((n, cout, 2 * h, 2 * wd), dtype=np.float64)
    x64 = x.astype(np.float64)
    for a in range(2):
        for bb in range(2):
            y[:, :, a::2, bb::2] = np.einsum("nchw,co->nohw", x64, w[:, :, a, bb].astype(np.float64))
    return (y + b.astype(np.float64)[None, :, None, None]).astype(np.float32)


KB = (np.array([[-1, 2, -1], [2, 0, 2], [-1, 2, -1]], dtype=np.float32) / 4.0)


def uniform_dropout(x, mask):
    """x*mask + KB(x)*(1-mask), channel 0 only, float32 arithmetic like the reference (unet.py:39-41)."""
    xp = reflect_pad1(x[:, :1])
    h, w = x.shape[2:]
    kb = np.zeros_like(x[:, :1], dtype=np.float32)
    for u in range(3):
        for v in range(3):
            if KB[u, v] != 0:
                kb = kb + xp[:, :, u:u + h, v:v + w] * KB[u, v]
    out = x.copy()
    out[:, :1] = x[:, :1] * mask + kb * (1 - mask)
    return out


def unet_forward(x, sd, nsteps, intermediates=None):
    """Full forward, numpy only (slow: use <= 64x64)."""
    t = {} if intermediates is None else intermediates
    cv = lambda n, v: conv3x3_reflect(v, sd[n + ".weight"], sd[n + ".bias"], relu=True)
    enc = {1: ("e21", "e22"), 2: ("e31", "e32"), 3: ("e41", "e42"), 4: ("e51", "e52")}
    dec = {4: ("upconv1", "d11", "d12", "xe42"), 3: ("upconv2", "d21", "d22", "xe32"),
           2: ("upconv3", "d31", "d32", "xe22"), 1: ("upconv4", "d41", "d42", "xe12")}
    t["xe11"] = cv("e11", x)
    cur = t["xe12"] = cv("e12", t["xe11"])
    for lvl in range(1, nsteps + 1):
        a, b = enc[lvl]
        t[f"xp{lvl}"], _ = maxpool2x2(cur)
        t["x" + a] = cv(a, t[f"xp{lvl}"])
        cur = t["x" + b] = cv(b, t["x" + a])
    for depth in range(nsteps, 0, -1):
        up, c1, c2, skip = dec[depth]
        xu = t["xu" + up[-1]] = convT2x2s2(cur, sd[up + ".weight"], sd[up + ".bias"])
        t["x" + c1] = cv(c1, np.concatenate([xu, t[skip]], axis=1))
        cur = t["x" + c2] = cv(c2, t["x" + c1])
    t["logit"] = conv1x1(cur, sd["outconv.weight"], sd["outconv.bias"])
    return sigmoid(t["logit"])


# ---------------------------------------------------------------------------
# evaluate-side statistics and losses
# ---------------------------------------------------------------------------

def ws_stats(x_u8: np.ndarray, x_hat: np.ndarray):
    """evaluate.py:125-132 on the cropped interior.  x_u8 (h,w) uint8-valued image
    (already cropped), x_hat (h,w) float32 prediction in 0..255 units.
    Returns (beta_hat, l1) as float64 exact means (the reference uses numpy's
    float32 pairwise mean; difference is ~1e-7 relative)."""
    x = x_u8.astype(np.float64)
    x_bar = (x_u8.astype(np.uint8) ^ 1).astype(np.float64)
    d = x - x_hat.astype(np.float64)
    return float(np.mean((x - x_bar) * d)), float(np.mean(np.abs(d)))


def lsb_flip_from_unit(x01: np.ndarray) -> np.ndarray:
    """losses.py:49-51: (round(x*255).int() ^ 1).float(), x in [0,1] float32.
    np.rint == torch.round (half to even)."""
    v = np.rint(x01.astype(np.float32) * np.float32(255.0)).astype(np.int32)
    return (v ^ 1).astype(np.float32)


def l1ws_loss(outputs, covers, alphas, inputs, use_l1=True, use_ws=True):
    """L1Loss + WSLoss (losses.py:33-36,47-89,99-116).  Returns (loss, dloss/doutputs) float64."""
    o = outputs.astype(np.float64)
    n = o.shape[0]
    numel = o.size
    loss = 0.0
    grad = np.zeros_like(o)
    if use_l1:
        d = covers.astype(np.float64) - o
        loss += np.mean(np.abs(d))
        grad += -np.sign(d) / numel
    if use_ws:
        in255 = inputs.astype(np.float32) * np.float32(255.0)
        out255 = o * 255.0
        bar = lsb_flip_from_unit(inputs).astype(np.float64)
        wgt = 1.0 / (numel / n)
        s = (in255.astype(np.float64) - bar)                       # +-1
        beta_hat = np.sum(wgt * s * (in255.astype(np.float64) - out255), axis=(1, 2, 3))
        pos = beta_hat > 0
        bh = np.where(pos, beta_hat, 0.0)
        e = bh - alphas.astype(np.float64) / 2.0
        loss += np.mean(np.abs(e))
        # d|e|/dout = sign(e) * relu'(beta_hat) * (-255 * wgt * s) / n
        coef = (np.sign(e) * pos / n)[:, None, None, None]
        grad += coef * (-255.0 * wgt) * s
    return float(loss), grad


def adamw_step(p, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8, wd=1e-2):
    """One torch.optim.AdamW update (decoupled weight decay), float64 math on float32 state."""
    p = p.astype(np.float64) * (1.0 - lr * wd)
    m = b1 * m.astype(np.float64) + (1 - b1) * g
    v = b2 * v.astype(np.float64) + (1 - b2) * g.astype(np.float64) ** 2
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p.astype(np.float32), m.astype(np.float32), v.astype(np.float32)
