"""GPU tests of mode 'f16f8' (f16 products + block-scaled fp8 cross terms, include/wsu.h): storage format, kernel chain and whole
networks against the split-bf16 path, the exact fp32 mode and the CPU oracle.  Tolerances: the mode carries ~2^-15 relative error per
product (bf16x3: ~2^-17); whole-network outputs stay 20x inside the 1e-4 MAE gate of BASELINE.json."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, images01, gpu_model, oracle_forward, unsplit_f16f8
from ws_unet_amd import formula, ops

pytestmark = pytest.mark.gpu
X3, F8 = ops.mode_id("bf16x3"), ops.mode_id("f16f8")


def _w(key, shape, scale):
    return torch.from_numpy(formula.formula_tensor(key, shape, scale)).to(DEV)


@pytest.mark.parametrize("hw", [(16, 32), (24, 40), (8, 8), (36, 70)])
def test_kernel_chain_against_split_bf16(hw):
    """fused first layer -> conv (+pool) -> conv -> transposed conv -> concat conv -> conv + head: every intermediate decodes (f16 part +
    residual) to the fp32-storage tensor of the bf16x3 chain within the mode's error, and the head output agrees to 1e-5."""
    h, w = hw
    x = images01(2, h, w, seed=3)[1].to(DEV)
    w1, b1 = _w(f"ps/w1/{hw}", (64, 1, 3, 3), 0.5), _w(f"ps/b1/{hw}", (64,), 0.1)
    w2, b2 = _w(f"ps/w2/{hw}", (64, 64, 3, 3), 0.06), _w(f"ps/b2/{hw}", (64,), 0.1)
    w3, b3 = _w(f"ps/w3/{hw}", (128, 64, 3, 3), 0.06), _w(f"ps/b3/{hw}", (128,), 0.1)
    wu, bu = _w(f"ps/wu/{hw}", (128, 64, 2, 2), 0.09), _w(f"ps/bu/{hw}", (64,), 0.1)
    w4, b4 = _w(f"ps/w4/{hw}", (64, 128, 3, 3), 0.04), _w(f"ps/b4/{hw}", (64,), 0.1)
    w5, b5 = _w(f"ps/w5/{hw}", (64, 64, 3, 3), 0.06), _w(f"ps/b5/{hw}", (64,), 0.1)
    hw_, hb = _w(f"ps/hw/{hw}", (1, 64, 1, 1), 0.3), _w(f"ps/hb/{hw}", (1,), 0.1)
    outs = {}
    for m in (X3, F8):
        y12, yp = ops.conv3x3_fused_first(x, w1, b1, ops.pack_conv3x3(w2, m), b2, 64, m, pool=True)
        y21 = ops.conv3x3(yp, None, ops.pack_conv3x3(w3, m), b3, 128, m)
        yu = ops.convt2x2(y21, ops.pack_convt2x2(wu, m), bu, 64, m)
        yd = ops.conv3x3(yu, y12, ops.pack_conv3x3(w4, m), b4, 64, m)
        out, logit = ops.conv3x3_head(yd, None, ops.pack_conv3x3(w5, m), b5, hw_, hb, m, want_logit=True)
        outs[m] = (y12, yp, y21, yu, yd, out, logit)
    for a, b in zip(outs[X3][:5], outs[F8][:5]):
        ref = a.cpu()
        got = unsplit_f16f8(b)
        assert b.shape[3] * 4 == ref.shape[3] * 3                      # 3 bytes per element
        scale = max(ref.abs().max().item(), 1e-30)
        assert (got - ref).abs().max().item() <= 2e-4 * scale
    assert (outs[X3][5] - outs[F8][5]).abs().max().item() <= 1e-5
    assert (outs[X3][6] - outs[F8][6]).abs().max().item() <= 1e-4 * max(outs[X3][6].abs().max().item(), 1.0)
    with pytest.raises(Exception, match="F16F8"):
        ops.conv3x3(outs[F8][0], None, ops.pack_conv3x3(w2, F8), b2, 64, F8, pool=True, pool_idx=True)


def test_stored_halves_are_the_documented_encoding():
    """f16 part + e4m3 residual * 2^-12 reproduces the layer's fp32 result to ~2^-16 and the residual stays within half an f16 ulp of its
    f16 part (it is a rounding residual, not a second value)."""
    x = images01(1, 16, 32, seed=5)[1].to(DEV)
    w1, b1 = _w("enc/w1", (64, 1, 3, 3), 0.5), _w("enc/b1", (64,), 0.1)
    w2, b2 = _w("enc/w2", (64, 64, 3, 3), 0.06), _w("enc/b2", (64,), 0.1)
    y = ops.conv3x3_fused_first(x, w1, b1, ops.pack_conv3x3(w2, F8), b2, 64, F8)
    yref = ops.conv3x3_fused_first(x, w1, b1, ops.pack_conv3x3(w2, X3), b2, 64, X3).cpu()
    raw = y.cpu().numpy().view(np.uint8).reshape(1, 16, 32, 4, 48)
    hi = raw[..., :32].copy().view(np.float16).astype(np.float32).reshape(1, 16, 32, 64)
    got = unsplit_f16f8(y)
    scale = yref.abs().max().item()
    assert (got - yref).abs().max().item() <= 2e-4 * scale                      # the two arithmetics agree to the mode's error
    half_ulp = np.maximum(np.abs(hi) * 2.0 ** -11, 2.0 ** -25)
    assert (np.abs(got.numpy() - hi) <= half_ulp * 1.07).all()


@pytest.mark.parametrize("ns", [1, 2, 3])
def test_whole_network(ns):
    _, x = images01(2, 64, 64, seed=13)
    m = gpu_model(ns, "he", "f16f8")
    with torch.no_grad():
        y = m(x.to(DEV))
        y32 = gpu_model(ns, "he", "f32")(x.to(DEV))
        keep = {}
        y3 = m.forward_features(x.to(DEV), keep=keep)              # keep= runs the bf16x3 path in fp32 storage
    d = (y - y32).abs()
    assert d.mean().item() <= 2e-5 and d.max().item() <= 1.5e-4      # measured 4e-6 / 4e-5 (unet_2), 9e-6 / 5.4e-5 (unet_3)
    assert "xe11" in keep and (y3 - y32).abs().max().item() <= 4e-5
    assert (y.cpu() - oracle_forward(x, ns)).abs().mean().item() <= 2e-5
    assert m.train_mode == "bf16x3"                                 # training is unaffected (fp32 storage)


def test_f16f8_512(golden):
    g = golden["unet_fwd_512"]
    _, x = images01(1, 512, 512, seed=7)
    with torch.no_grad():
        y = gpu_model(2, "he", "f16f8")(x.to(DEV))
    crop = y[0, 0].cpu().numpy()[224:288, 224:288]
    d = np.abs(crop - g["f512_he_crop"])
    assert d.mean() <= 2e-5 and d.max() <= 1.5e-4


def test_large_values_degrade_gracefully():
    """Activations beyond e4m3's scaled range (|x| > 448) lose only the residual term: the result keeps plain-f16 accuracy (2^-11
    relative) instead of overflowing to NaN."""
    h, w = 16, 32
    x = (images01(1, h, w, seed=2)[1] * 3000.0).to(DEV)           # first-layer outputs in the thousands
    w1, b1 = _w("big/w1", (64, 1, 3, 3), 0.5), _w("big/b1", (64,), 0.1)
    w2, b2 = _w("big/w2", (64, 64, 3, 3), 0.06), _w("big/b2", (64,), 0.1)
    w3, b3 = _w("big/w3", (64, 64, 3, 3), 0.06), _w("big/b3", (64,), 0.1)
    res = {}
    for m in (X3, F8):
        y = ops.conv3x3_fused_first(x, w1, b1, ops.pack_conv3x3(w2, m), b2, 64, m)
        res[m] = ops.conv3x3(y, None, ops.pack_conv3x3(w3, m), b3, 64, m)
    ref = res[X3].cpu()
    got = unsplit_f16f8(res[F8])
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 2.0 ** -9 * ref.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 16, 32, 64, 64), (1, 24, 40, 128, 64), (1, 8, 8, 64, 128)])
def test_f16f8x_on_fp32_tensors(shape):
    """Mode 'f16f8x' (the training forward): the f16f8 arithmetic on fp32 tensors, with the fused pool and its argmax, against the
    split-bf16 kernels on the same tensors."""
    n, h, w, cin, cout = shape
    FX = ops.mode_id("f16f8x")
    x = torch.from_numpy(formula.formula_tensor(f"fx/x/{shape}", (n, h, w, cin), 1.0)).abs().to(DEV)
    wt, b = _w(f"fx/w/{shape}", (cout, cin, 3, 3), 0.05), _w(f"fx/b/{shape}", (cout,), 0.1)
    y1, p1, i1 = ops.conv3x3(x, None, ops.pack_conv3x3(wt, X3), b, cout, X3, pool=True, pool_idx=True)
    y2, p2, i2 = ops.conv3x3(x, None, ops.pack_conv3x3(wt, FX), b, cout, FX, pool=True, pool_idx=True)
    scale = y1.abs().max().item()
    assert y2.dtype == torch.float32 and y2.shape == y1.shape
    assert (y1 - y2).abs().max().item() <= 1e-4 * scale and (p1 - p2).abs().max().item() <= 1e-4 * scale
    # the argmax names an element of the window that carries the pooled value (ties / near-ties may pick another position)
    win = y2.reshape(n, h // 2, 2, w // 2, 2, cout).permute(0, 1, 3, 5, 2, 4).reshape(n, h // 2, w // 2, cout, 4)
    assert torch.equal(torch.gather(win, 4, i2.long().unsqueeze(-1)).squeeze(-1), p2)
    assert (i1 != i2).float().mean().item() <= 0.01
    # concat input and the transposed conv
    x2 = torch.from_numpy(formula.formula_tensor(f"fx/x2/{shape}", (n, h, w, 64), 1.0)).to(DEV)
    wc = _w(f"fx/wc/{shape}", (64, cin + 64, 3, 3), 0.04)
    c1 = ops.conv3x3(x, x2, ops.pack_conv3x3(wc, X3), None, 64, X3)
    c2 = ops.conv3x3(x, x2, ops.pack_conv3x3(wc, FX), None, 64, FX)
    assert (c1 - c2).abs().max().item() <= 1e-4 * c1.abs().max().item()
    wu, bu = _w(f"fx/wu/{shape}", (cin, 64, 2, 2), 0.09), _w(f"fx/bu/{shape}", (64,), 0.1)
    u1 = ops.convt2x2(x, ops.pack_convt2x2(wu, X3), bu, 64, X3)
    u2 = ops.convt2x2(x, ops.pack_convt2x2(wu, FX), bu, 64, FX)
    assert u2.shape == (n, 2 * h, 2 * w, 64) and (u1 - u2).abs().max().item() <= 1e-4 * u1.abs().max().item()
    with pytest.raises(ValueError, match="training forward"):
        gpu_model(2, "he", "f16f8x")


@pytest.mark.parametrize("shape", [(2, 16, 32, 128, 64), (1, 12, 40, 128, 128)])
def test_f16f8x_data_gradient(shape):
    """The data gradient of the reflect-padded conv (interior conv with flipped weights, border-ring fold, ReLU masks, concat split) in
    the f16f8 arithmetic against the split-bf16 one, on gradients brought into f16's range the way the autograd node does."""
    n, h, w, cin, cout = shape
    FX = ops.mode_id("f16f8x")
    g = torch.from_numpy(formula.formula_tensor(f"dg/g/{shape}", (n, h, w, cout), 1.0)).to(DEV) * 3e-7       # a mean-reduced loss's scale
    wt = _w(f"dg/w/{shape}", (cout, cin, 3, 3), 0.05)
    mask = torch.from_numpy(formula.formula_tensor(f"dg/m/{shape}", (n, h, w, cin), 1.0)).to(DEV)
    m1, m2 = mask[..., :cin // 2].contiguous(), mask[..., cin // 2:].contiguous()
    scale = torch.exp2(torch.floor(2.0 - torch.log2(g.abs().max())))
    a1, a2 = ops.conv3x3_bwd_data(g, ops.pack_conv3x3(wt, X3, dgrad=True), wt, cin // 2, m1, m2, X3)
    b1, b2 = ops.conv3x3_bwd_data(g * scale, ops.pack_conv3x3(wt, FX, dgrad=True), wt, cin // 2, m1, m2, FX)
    for a, b in ((a1, b1), (a2, b2)):
        b = b / scale
        assert torch.equal(a == 0, b == 0) or ((a == 0) != (b == 0)).float().mean().item() < 1e-3        # the ReLU masks zero the same places
        assert (a - b).abs().max().item() <= 1e-4 * a.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 16, 32, 128, 64), (1, 12, 40, 64, 128), (2, 6, 70, 64, 64)])
def test_f16f8x_weight_gradient(shape):
    """dW / db of the 3x3 conv (concat input) and of the transposed conv in the f16f8 arithmetic (f16 products per 16 pixels, both cross
    terms of 32 pixels in one block-scaled fp8 MFMA, transposing 8-bit LDS reads) against the split-bf16 kernels, on a gradient
    brought into f16's range the way the autograd node does."""
    n, h, w, cin, cout = shape
    FX = ops.mode_id("f16f8x")
    g = torch.from_numpy(formula.formula_tensor(f"wg/g/{shape}", (n, h, w, cout), 1.0)).to(DEV) * 3e-7
    x1 = torch.from_numpy(formula.formula_tensor(f"wg/x1/{shape}", (n, h, w, cin), 1.0)).abs().to(DEV)
    x2 = torch.from_numpy(formula.formula_tensor(f"wg/x2/{shape}", (n, h, w, 64), 1.0)).to(DEV)
    scale = torch.exp2(torch.floor(2.0 - torch.log2(g.abs().max())))
    dw1, db1 = ops.conv3x3_bwd_weight(g, x1, x2, mode=X3)
    dw2, db2 = ops.conv3x3_bwd_weight(g * scale, x1, x2, mode=FX)
    dw2, db2 = dw2 / scale, db2 / scale
    assert (dw1 - dw2).abs().max().item() <= 2e-4 * dw1.abs().max().item()
    assert (db1 - db2).abs().max().item() <= 1e-6 * db1.abs().max().item() + 1e-12        # exact fp32 sums on both sides
    # transposed conv: U = the low-resolution activation, V = the (scaled) gradient at twice the resolution
    xl = torch.from_numpy(formula.formula_tensor(f"wg/xl/{shape}", (n, h, w, cin), 1.0)).to(DEV)
    dy = torch.from_numpy(formula.formula_tensor(f"wg/dy/{shape}", (n, 2 * h, 2 * w, cout), 1.0)).to(DEV) * 3e-7
    s2 = torch.exp2(torch.floor(2.0 - torch.log2(dy.abs().max())))
    tw1, tb1 = ops.convt2x2_bwd_weight(xl, dy, mode=X3)
    tw2, tb2 = ops.convt2x2_bwd_weight(xl, dy * s2, mode=FX)
    assert (tw1 - tw2 / s2).abs().max().item() <= 2e-4 * tw1.abs().max().item()
    assert (tb1 - tb2 / s2).abs().max().item() <= 1e-5 * tb1.abs().max().item() + 1e-12
