"""Backward parity at PRODUCTION sizes: the size-dependent paths of wgrad.hip (split-K slab counts, the
column-major tile walk with its rolling row window, adaptive chunking of the bias / first-layer reductions) and backward.hip (border
strips of width W+2, ring fold) only run at 256^2 ... 1024^2.  Every kernel, in every training arithmetic ('f32', 'bf16x3', 'f16f8x'),
is compared with torch autograd in fp32 on the CPU oracle -- an independent implementation, not another libwsu kernel -- by relative
L2 error per tensor (tolerances from the fp64 error tables in profiles/r01/grad_error_vs_fp64_*.txt: 2^-16 per product for the split
modes -> ~1e-5 relative L2 on a random-sign reduction, plus the fp32 oracle's own ~1e-6)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import DEV, to_nhwc, from_nhwc, rand_act, gpu_model
from ws_unet_amd import formula, ops, losses
from oracle import unet_ref, losses_ref

pytestmark = pytest.mark.gpu

# relative L2 per tensor against the fp32 CPU oracle
REL_L2 = {"f32": 5e-6, "bf16x3": 1.5e-5, "f16f8x": 3e-5}        # measured on MI355X (r02): 0.5-1.4e-6 / 3.7-4.2e-6 / 8.5-9.5e-6
DB_TOL = 2e-5      # bias gradients: fp32 sums of 1e5..5e5 random-sign values, the summation order alone moves them by ~2-6e-6
MODES = ["f32", "bf16x3", "f16f8x"]


def rel_l2(got: torch.Tensor, ref: torch.Tensor) -> float:
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-300))


def check(got, ref, tol, what, log):
    e = rel_l2(got, ref)
    log.append(f"{what}: rel L2 {e:.2e} (tol {tol:.0e})")
    assert e <= tol, "; ".join(log)


def check_trimmed(got, ref, tol, what, log, trim=2e-3):
    """Relative L2 after dropping the `trim` fraction of elements with the largest deviation.  For the INPUT gradient: a single ReLU
    mask that flips on fp32 rounding noise (a pre-activation within ~1e-7 of zero: a few per 10^8 activations) changes dL/dx by O(100 %)
    in its 3x3 neighbourhood, and dL/dx is not averaged over pixels the way a weight gradient is (tools/diag_dx.py: the kernel alone
    agrees to 2.6e-7; the whole-net difference sits in isolated columns).  Everything outside those few neighbourhoods must agree."""
    d = (got.double().cpu() - ref.double().cpu()).abs().reshape(-1)
    k = int(d.numel() * (1.0 - trim))
    kept = torch.topk(d, k, largest=False).values
    e = float(kept.norm() / ref.double().norm().clamp_min(1e-300))
    log.append(f"{what}: trimmed rel L2 {e:.2e} (tol {tol:.0e}; untrimmed {rel_l2(got, ref):.2e})")
    assert e <= tol, "; ".join(log)


_oracle_cache = {}


def conv_oracle(shape):
    """torch-CPU fp32 autograd of relu-free conv3x3-reflect at a production layer shape (computed once per shape, ~2-4 s)."""
    if shape in _oracle_cache:
        return _oracle_cache[shape]
    n, h, w, c1, c2, cout = shape
    x1 = rand_act((n, c1, h, w), f"L/x1/{shape}").requires_grad_(True)
    x2 = rand_act((n, c2, h, w), f"L/x2/{shape}").requires_grad_(True) if c2 else None
    wt = torch.from_numpy(formula.formula_tensor(f"L/w/{shape}", (cout, c1 + c2, 3, 3), (6.0 / (9 * (c1 + c2))) ** 0.5)).requires_grad_(True)
    b = torch.from_numpy(formula.formula_tensor(f"L/b/{shape}", (cout,), 0.1)).requires_grad_(True)
    g = torch.from_numpy(formula.formula_tensor(f"L/g/{shape}", (n, cout, h, w), 1.0))
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    unet_ref.conv3x3_reflect(xin, wt, b).backward(g)
    res = dict(x1=x1.detach(), x2=None if x2 is None else x2.detach(), w=wt.detach(), g=g,
               dw=wt.grad, db=b.grad, dx1=x1.grad, dx2=None if x2 is None else x2.grad)
    _oracle_cache.clear()                   # keep one shape resident (hundreds of MB each)
    _oracle_cache[shape] = res
    return res


# (N, H, W, C1, C2, Cout): e12/d42 at full resolution (split-K over 2048 pixel tiles), d31 with the fused concat at 256^2, and a tall
# thin image whose every split owns several tiles of one column (rolling row window of the weight-gradient walk)
SHAPES = [(2, 512, 512, 64, 0, 64), (2, 256, 256, 128, 128, 128), (1, 1024, 64, 64, 0, 64)]


@pytest.mark.parametrize("shape", SHAPES)
def test_conv3x3_gradients_production_size(shape):
    o = conv_oracle(shape)
    n, h, w, c1, c2, cout = shape
    gd = to_nhwc(o["g"], "f32")
    x1d = to_nhwc(o["x1"], "f32")
    x2d = None if o["x2"] is None else to_nhwc(o["x2"], "f32")
    wd = o["w"].to(DEV)
    log = []
    for mode in MODES:
        m = ops.mode_id(mode)
        tol = REL_L2[mode]
        dw, db = ops.conv3x3_bwd_weight(gd, x1d, x2d, mode=m)
        check(dw, o["dw"], tol, f"{mode} dW", log)
        check(db, o["db"], DB_TOL, f"{mode} db", log)                     # exact fp32 sums in every mode
        dx1, dx2 = ops.conv3x3_bwd_data(gd, ops.pack_conv3x3(wd, m, dgrad=True), wd, c1, x1d, None, m)
        check(from_nhwc(dx1), o["dx1"] * (o["x1"] > 0), tol, f"{mode} dx1 (ReLU-masked)", log)
        if c2:
            check(from_nhwc(dx2), o["dx2"], tol, f"{mode} dx2", log)
        # border rows / columns alone (the reflect adjoint: ring strips + fold) must meet the same band
        ring = from_nhwc(dx1)
        ref = o["dx1"] * (o["x1"] > 0)
        for sl in ((slice(None), slice(None), slice(0, 2)), (slice(None), slice(None), slice(h - 2, h)),
                   (slice(None), slice(None), slice(None), slice(0, 2)), (slice(None), slice(None), slice(None), slice(w - 2, w))):
            check(ring[sl], ref[sl], 2 * tol, f"{mode} dx1 border {sl[2:]}", log)
        del dw, db, dx1, dx2
    print("\n".join(log))


@pytest.mark.parametrize("shape", [(2, 128, 128, 256, 128), (1, 256, 256, 128, 64)])     # upconv3 / upconv4 of unet_2 at 512^2
def test_convt2x2_gradients_production_size(shape):
    n, h, w, cin, cout = shape
    x = rand_act((n, cin, h, w), f"LT/x/{shape}").requires_grad_(True)
    wt = torch.from_numpy(formula.formula_tensor(f"LT/w/{shape}", (cin, cout, 2, 2), (6.0 / cin) ** 0.5)).requires_grad_(True)
    b = torch.from_numpy(formula.formula_tensor(f"LT/b/{shape}", (cout,), 0.1)).requires_grad_(True)
    dy = torch.from_numpy(formula.formula_tensor(f"LT/dy/{shape}", (n, cout, 2 * h, 2 * w), 1.0))
    F.conv_transpose2d(x, wt, b, stride=2).backward(dy)
    dyd, xd = to_nhwc(dy, "f32"), to_nhwc(x.detach(), "f32")
    log = []
    for mode in MODES:
        m = ops.mode_id(mode)
        dw, db = ops.convt2x2_bwd_weight(xd, dyd, mode=m)
        check(dw, wt.grad, REL_L2[mode], f"{mode} convT dW", log)
        check(db, b.grad, DB_TOL, f"{mode} convT db", log)
        md = ops.mode_id("bf16x3") if mode == "f16f8x" else m                   # the autograd node runs this kernel in 'bf16x3' (autograd.py)
        dx = ops.convt2x2_bwd_data(dyd, ops.pack_convt2x2_dgrad(wt.detach().to(DEV), md), cin, xd, md)
        check(from_nhwc(dx), x.grad * (x.detach() > 0), REL_L2["bf16x3" if mode == "f16f8x" else mode], f"{mode} convT dx", log)
    print("\n".join(log))


def _pairs(n, size, seed):
    cov_u8 = formula.synthetic_images(n, size, size, seed=seed)
    st_u8 = np.stack([formula.lsbr_embed(c, 0.4, seed=seed + i) if i % 2 == 0 else c for i, c in enumerate(cov_u8)])
    covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None]
    inputs = torch.from_numpy(st_u8.astype(np.float32) / np.float32(255.))[:, None]
    alphas = torch.tensor([0.4 if i % 2 == 0 else 0.0 for i in range(n)])
    return covers, inputs, alphas


def _oracle_step(size, n, seed, loss_name):
    covers, inputs, alphas = _pairs(n, size, seed)
    ref = unet_ref.build_ref(2, formula.formula_state_dict(2, "he"))
    x = inputs.clone().requires_grad_(True)
    out = ref(x)
    loss = losses_ref.LOSSES[loss_name](out, (covers, alphas), x.detach())
    loss.backward()
    return covers, inputs, alphas, out.detach(), loss.item(), {k: p.grad.clone() for k, p in ref.named_parameters()}, x.grad.clone()


# per-tensor relative L2 bands: (f32 model, default split-bf16 model with the f16f8x arithmetic)
#   'l2'   smooth loss: only ReLU-mask flips separate two implementations -> tight band (measured r02: see gpurun_out/r2_pytest_*.log)
#   'l1ws' sign(cover - out) flips wherever |cover - out| is below the forward's rounding noise (2.5e-6 in fp32): k flipped pixels of
#          N*H*W move dL/dout by sqrt(4k / NHW) relative L2 -- 3.5e-3 for k = 3 at 1024^2 -- for ANY two fp32 implementations
#   third entry: the DEFAULT training arithmetic, train_mode 'f16f8p' (planar activations and gradients, 3 B per element: every saved
#          activation carries the format's 2^-15 relative rounding, every gradient plane its own) -- the band is the measured level of
#          profiles/r02/grad_error_vs_fp64_f16f8p_then_bf16x3.txt (3-7e-4 per parameter on 64x64 under L1WS) plus the mask-flip floor above
GRAD_TOL = {"l2": (3e-4, 1e-3, 2.5e-3), "l1ws": (8e-3, 1.2e-2, 1.5e-2)}


@pytest.mark.parametrize("loss_name", ["l2", "l1ws"])
@pytest.mark.parametrize("size,n", [(1024, 1), (512, 2)])
def test_unet2_forward_backward_large_vs_oracle(size, n, loss_name):
    """BASELINE.json configs[4] runs 1024x1024 pairs: one whole unet_2 forward + loss + backward at that size (and a batch of two at
    512^2) against the CPU oracle's autograd, in exact fp32, in the split-bf16 model's training arithmetic (f16f8x forward, data and weight
    gradients on fp32 tensors) and in the DEFAULT one (train_mode 'f16f8p': planar activations and gradients -- what bench.py's train_step
    times), by relative L2 per tensor (the small-size golden test allows 1.5e-2 x max)."""
    covers, inputs, alphas, out_ref, loss_ref, grads_ref, dx_ref = _oracle_step(size, n, 300 + size, loss_name)
    log = []
    crit = {"l2": losses.L2Loss, "l1ws": losses.L1WSLoss}[loss_name]
    for (mode, tol_out), tol_g in zip((("f32", 4e-6), ("bf16x3", 1e-4), ("f16f8p", 1e-4)), GRAD_TOL[loss_name]):
        model = gpu_model(2, "he", mode)
        planar = mode == "f16f8p"
        # the planar path has no input-gradient kernel (a model asked for dL/dx takes the fp32-storage path for that call,
        # test_gpu_planar_train.py::test_planar_training_fallback_for_input_gradients): the default arithmetic is run as the trainer runs it
        x = inputs.to(DEV).requires_grad_(not planar)
        if planar:
            assert model.train_mode == "f16f8p"
            timer = ops.KernelTimer()
            ops.set_timer(timer)
        out = model(x)
        loss = crit()(out, (covers.to(DEV), alphas.to(DEV)), x)
        loss.backward()
        if planar:
            torch.cuda.synchronize()
            ops.set_timer(None)
            used = timer.summary()
            assert "conv3x3_pl_bwd_data" in used and "conv3x3_pl_bwd_weight" in used and "conv3x3_bwd_data" not in used, sorted(used)
            assert not model.range_exceeded()
        assert math.isclose(loss.item(), loss_ref, rel_tol=2e-5 if mode == "f32" else 2e-4), (mode, loss.item(), loss_ref)
        err = (out.detach().cpu() - out_ref).abs()
        log.append(f"{mode} out: max {err.max():.2e} mean {err.mean():.2e}")
        assert err.max().item() <= (tol_out if mode == "f32" else 2e-4) and err.mean().item() <= tol_out, "; ".join(log)
        for k, p in model.named_parameters():
            check(p.grad, grads_ref[k], tol_g, f"{mode} {k}", log)
        # a flip in a deep layer reaches a ~45 x 45-pixel receptive field of dL/dx: trim 1 % and allow 10 x the weight-gradient band
        if not planar:
            check_trimmed(x.grad, dx_ref, 10 * tol_g, f"{mode} dL/dx (network path)", log, trim=1e-2)
        del model, out, loss, x
        torch.cuda.empty_cache()
    print("\n".join(log))


@pytest.mark.parametrize("ns", [0, 1, 2])
def test_input_gradient_golden(golden, ns):
    """d loss / d x THROUGH the network against the reference's own autograd (tests/golden/make_golden.py: grad{ns}_dx_net, the loss
    fed a detached copy of the inputs -- the fused loss kernel treats the inputs as data, like the training loop)."""
    g = golden["unet_grad"]
    model = gpu_model(ns, "he", "f32")
    cov_u8 = formula.synthetic_images(2, 64, 64, seed=11)
    st_u8 = cov_u8.copy(); st_u8[0] = formula.lsbr_embed(cov_u8[0], 0.4, seed=5)
    covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None].to(DEV)
    inputs = torch.from_numpy(st_u8.astype(np.float32) / np.float32(255.))[:, None].to(DEV).requires_grad_(True)
    losses.L1WSLoss()(model(inputs), (covers, torch.tensor([0.4, 0.0], device=DEV)), inputs).backward()
    ref = torch.from_numpy(g[f"grad{ns}_dx_net"])
    log = []
    check_trimmed(inputs.grad, ref, 1e-2, f"unet_{ns} dL/dx", log, trim=1e-2)       # L1's sign flips on top of the ReLU flips


@pytest.mark.parametrize("fwd,bwd", [("f16f8x", "f16f8x"), ("bf16x3", "bf16x3"), ("f16f8x", "bf16x3")])
def test_unet_gradients_golden_per_train_arithmetic(golden, fwd, bwd):
    """Both arithmetics the WSU_TRAIN_FWD_MODE / WSU_TRAIN_BWD_MODE switches select, end to end against the reference goldens by
    relative L2 per tensor (the pure split-bf16 path had no end-to-end check after 'f16f8x' became the default)."""
    g = golden["unet_grad"]
    model = gpu_model(2, "he", "bf16x3")
    model.train_fwd_mode, model.train_bwd_mode = fwd, bwd
    cov_u8 = formula.synthetic_images(2, 64, 64, seed=11)
    st_u8 = cov_u8.copy(); st_u8[0] = formula.lsbr_embed(cov_u8[0], 0.4, seed=5)
    covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None].to(DEV)
    inputs = torch.from_numpy(st_u8.astype(np.float32) / np.float32(255.))[:, None].to(DEV)
    out = model(inputs)
    losses.L1WSLoss()(out, (covers, torch.tensor([0.4, 0.0], device=DEV)), inputs).backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["grad2_out"], atol=1e-4, rtol=0)
    for k, p in model.named_parameters():
        full = p.grad.detach().double().cpu().numpy()
        assert math.isclose(float(np.sqrt((full ** 2).sum())), g[f"grad2_{k}_sum"][2], rel_tol=8e-3), (fwd, bwd, k)
        got = full.reshape(-1)
        ref = g[f"grad2_{k}_sub"].astype(np.float64)
        if got.size > 4096:
            got = got[::97]
        e = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-300)
        assert e <= 1e-2, f"{fwd}/{bwd} {k}: rel L2 {e:.2e}"


def test_disable_center_pixels():
    """unet.py:196-199: e11.weight[:, :, 1, 1] and its gradient are zeroed; the packed-weight cache must notice, so the forward
    changes accordingly and equals the oracle run on the edited weights."""
    model = gpu_model(1, "he", "f32")
    x = torch.from_numpy(formula.synthetic_images(1, 32, 32, seed=7).astype(np.float32) / np.float32(255.))[:, None].to(DEV)
    with torch.no_grad():
        before = model(x).clone()
    # populate e11.weight.grad, then zero the centre tap
    losses.L1Loss()(model(x), (torch.zeros_like(x), torch.zeros(1, device=DEV))).backward()
    assert model.e11.weight.grad[:, :, 1, 1].abs().sum().item() > 0
    others_w = model.e11.weight.detach().clone()
    others_g = model.e11.weight.grad.clone()
    model.disable_center_pixels()
    assert torch.count_nonzero(model.e11.weight[:, :, 1, 1]).item() == 0
    assert torch.count_nonzero(model.e11.weight.grad[:, :, 1, 1]).item() == 0
    keep = torch.ones(3, 3, dtype=torch.bool); keep[1, 1] = False
    assert torch.equal(model.e11.weight.detach()[:, :, keep], others_w[:, :, keep])          # nothing else touched
    assert torch.equal(model.e11.weight.grad[:, :, keep], others_g[:, :, keep])
    with torch.no_grad():
        after = model(x)
    assert (after - before).abs().max().item() > 1e-4                                       # stale packed weights would give `before`
    sd = formula.formula_state_dict(1, "he")
    sd["e11.weight"] = sd["e11.weight"].copy(); sd["e11.weight"][:, :, 1, 1] = 0
    ref = unet_ref.unet_forward(x.cpu().clone(), unet_ref.to_torch_state(sd), 1)
    np.testing.assert_allclose(after.cpu().numpy(), ref.detach().numpy(), atol=4e-6, rtol=0)
    # the inference-format path (fused first layer) sees the edit too
    m2 = gpu_model(1, "he", "f16f8")
    m2.disable_center_pixels()
    with torch.no_grad():
        np.testing.assert_allclose(m2(x).cpu().numpy(), ref.detach().numpy(), atol=1e-4, rtol=0)
