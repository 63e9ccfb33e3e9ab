"""Pin the CPU oracle (oracle/) against golden vectors produced by the reference
itself (tests/golden/make_golden.py).  CPU only."""
import json
import math

import numpy as np
import pytest
import torch

from oracle import unet_ref, np_ops, losses_ref, metrics_ref
from ws_unet_amd import formula
from conftest import GOLDEN


def images01(n, h, w, seed):
    u8 = formula.synthetic_images(n, h, w, seed)
    return u8, torch.from_numpy(u8.astype(np.float32) / np.float32(255.))[:, None]


def test_formula_is_stable():
    # values pinned so a change of the generator is caught before it silently
    # invalidates every golden file
    t = formula.formula_tensor("probe", (4,), 1.0)
    assert t.dtype == np.float32
    np.testing.assert_array_equal(t, formula.formula_tensor("probe", (4,), 1.0))
    sd = formula.formula_state_dict(2, "he")
    assert sum(v.size for v in sd.values()) == 1861697
    assert list(sd)[:4] == ["e11.weight", "e11.bias", "e12.weight", "e12.bias"]
    assert list(sd)[-2:] == ["outconv.weight", "outconv.bias"]
    img = formula.synthetic_images(1, 8, 8, seed=1)
    assert img.dtype == np.uint8 and img.shape == (1, 8, 8)
    cov = formula.synthetic_images(1, 256, 256, seed=5)
    st = formula.lsbr_embed(cov, 0.4)
    assert np.all((cov ^ st) <= 1)                     # LSB-only change
    assert abs(np.mean(cov != st) - 0.2) < 0.01        # change rate alpha/2


@pytest.mark.parametrize("ns", [0, 1, 2, 3, 4])
def test_forward_small_torch_oracle(golden, ns):
    g = golden["unet_fwd_small"]
    sd = unet_ref.to_torch_state(formula.formula_state_dict(ns, "he"))
    _, x = images01(2, 32, 32, seed=1)
    with torch.no_grad():
        y = unet_ref.unet_forward(x.clone(), sd, ns)
    np.testing.assert_allclose(y.numpy(), g[f"y_unet{ns}_he"], rtol=0, atol=2e-7)


def test_forward_default_variant_and_odd_shape(golden):
    g = golden["unet_fwd_small"]
    _, x = images01(2, 32, 32, seed=1)
    sd = unet_ref.to_torch_state(formula.formula_state_dict(2, "default"))
    with torch.no_grad():
        y = unet_ref.unet_forward(x.clone(), sd, 2)
    np.testing.assert_allclose(y.numpy(), g["y_unet2_default"], rtol=0, atol=2e-7)
    sd = unet_ref.to_torch_state(formula.formula_state_dict(2, "he"))
    _, x = images01(2, 24, 40, seed=3)
    with torch.no_grad():
        y = unet_ref.unet_forward(x.clone(), sd, 2)
    np.testing.assert_allclose(y.numpy(), g["y_unet2_he_24x40"], rtol=0, atol=2e-7)


def test_intermediates_both_oracles(golden):
    g = golden["unet_fwd_small"]
    sd_np = formula.formula_state_dict(2, "he")
    _, x = images01(1, 32, 32, seed=2)
    t_t, t_n = {}, {}
    with torch.no_grad():
        y_t = unet_ref.unet_forward(x.clone(), unet_ref.to_torch_state(sd_np), 2, intermediates=t_t)
    y_n = np_ops.unet_forward(x.numpy(), sd_np, 2, intermediates=t_n)
    np.testing.assert_allclose(y_t.numpy(), g["inter_y"], atol=2e-7, rtol=0)
    np.testing.assert_allclose(y_n, g["inter_y"], atol=2e-6, rtol=0)
    names = [k[len("inter_"):-len("_sub")] for k in g.files if k.endswith("_sub")]
    assert set(names) == {"xe11", "xe12", "xp1", "xe21", "xe22", "xp2", "xe31", "xe32", "xu3", "xd31", "xd32",
                          "xu4", "xd41", "xd42", "logit"}
    for k in names:
        ref_sub = g[f"inter_{k}_sub"]
        scale = max(1.0, float(np.abs(ref_sub).max()))
        np.testing.assert_allclose(t_t[k].numpy()[:, ::8], ref_sub, atol=2e-6 * scale, rtol=0, err_msg=k)
        np.testing.assert_allclose(t_n[k][:, ::8], ref_sub, atol=2e-5 * scale, rtol=0, err_msg=k)
        s = g[f"inter_{k}_sum"]
        assert math.isclose(float(t_t[k].double().abs().sum()), s[1], rel_tol=1e-6), k


@pytest.mark.parametrize("variant", ["he", "default"])
def test_forward_512(golden, variant):
    g = golden["unet_fwd_512"]
    sd = unet_ref.to_torch_state(formula.formula_state_dict(2, variant))
    _, x = images01(1, 512, 512, seed=7)
    with torch.no_grad():
        y = unet_ref.unet_forward(x, sd, 2).numpy()[0, 0]
    np.testing.assert_allclose(y[224:288, 224:288], g[f"f512_{variant}_crop"], atol=5e-7, rtol=0)
    np.testing.assert_allclose(np.stack([y[0], y[511], y[:, 0], y[:, 511]]), g[f"f512_{variant}_border"], atol=5e-7, rtol=0)
    ts = y.astype(np.float64).reshape(8, 64, 8, 64).sum(axis=(1, 3))
    np.testing.assert_allclose(ts, g[f"f512_{variant}_tilesum"], rtol=1e-6)


@pytest.mark.parametrize("ns", [0, 1, 2])
def test_gradients_autograd_oracle(golden, ns):
    g = golden["unet_grad"]
    m = unet_ref.build_ref(ns, formula.formula_state_dict(ns, "he"))
    cov_u8 = formula.synthetic_images(2, 64, 64, seed=11)
    st_u8 = cov_u8.copy(); st_u8[0] = formula.lsbr_embed(cov_u8[0], 0.4, seed=5)
    covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None]
    inputs = torch.from_numpy(st_u8.astype(np.float32) / np.float32(255.))[:, None].requires_grad_(True)
    alphas = torch.tensor([0.4, 0.0])
    out = m(inputs)
    loss = losses_ref.l1ws_loss(out, (covers, alphas), inputs)
    loss.backward()
    assert math.isclose(loss.item(), float(g[f"grad{ns}_loss"][0]), rel_tol=1e-6)
    np.testing.assert_allclose(inputs.grad.numpy(), g[f"grad{ns}_dx"], rtol=1e-4, atol=1e-9)
    for k, p in m.named_parameters():
        gg = p.grad.numpy().reshape(-1)
        ref = g[f"grad{ns}_{k}_sub"]
        got = gg if (ns == 0 or gg.size <= 4096) else gg[::97]
        tol = 1e-5 * max(1e-12, float(np.abs(ref).max()))
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=tol, err_msg=k)


def test_dropout_external_mask(golden):
    g = golden["dropout"]
    _, x = images01(2, 32, 32, seed=21)
    mask = formula.bernoulli_mask((2, 1, 32, 32), keep_prob=0.9, seed=4242)
    xin = x.clone()
    xd = unet_ref.uniform_dropout(xin, torch.from_numpy(mask))
    np.testing.assert_allclose(xd.numpy(), g["drop_x_after"], atol=1e-7, rtol=0)
    assert xd.data_ptr() == xin.data_ptr()                       # in-place like the reference
    np.testing.assert_allclose(np_ops.uniform_dropout(x.numpy(), mask), g["drop_x_after"], atol=1e-6, rtol=0)
    sd = unet_ref.to_torch_state(formula.formula_state_dict(1, "he"))
    with torch.no_grad():
        y = unet_ref.unet_forward(x.clone(), sd, 1, dropout_mask=torch.from_numpy(mask))
        y0 = unet_ref.unet_forward(x.clone(), sd, 1, dropout_mask=torch.ones(2, 1, 32, 32))
    np.testing.assert_allclose(y.numpy(), g["drop_y_unet1"], atol=2e-7, rtol=0)
    np.testing.assert_allclose(y0.numpy(), g["drop0_y_unet1"], atol=2e-7, rtol=0)


def test_micro_vectors(golden):
    g = golden["micro"]
    w = formula.formula_tensor("micro/conv.w", (2, 1, 3, 3), 1.0)
    b = formula.formula_tensor("micro/conv.b", (2,), 1.0)
    x = np.arange(20, dtype=np.float32).reshape(1, 1, 4, 5)
    np.testing.assert_allclose(np_ops.conv3x3_reflect(x, w, b), g["micro_conv_y"], rtol=1e-6, atol=1e-6)
    x2 = np.arange(4, dtype=np.float32).reshape(1, 1, 2, 2)
    np.testing.assert_allclose(np_ops.conv3x3_reflect(x2, w, b), g["micro_conv_y_2x2"], rtol=1e-6, atol=1e-6)
    assert list(np_ops.reflect_index(np.arange(-1, 5), 4)) == [1, 0, 1, 2, 3, 2]
    xt = np.array([[[[0., 0., 1., 1.], [0., 0., 1., 2.], [3., 3., 0., 5.], [3., 1., 5., 5.]]]], dtype=np.float32)
    y, arg = np_ops.maxpool2x2(np.maximum(xt, 0))
    np.testing.assert_array_equal(y, g["micro_pool_y"])
    dy = np.array([[[[1., 2.], [3., 4.]]]], dtype=np.float32)
    dx = np_ops.maxpool2x2_backward(dy, arg, xt.shape) * (xt > 0)
    np.testing.assert_array_equal(dx, g["micro_pool_dx"])           # first-max-wins + relu'(0)=0
    wt = formula.formula_tensor("micro/convt.w", (3, 2, 2, 2), 1.0)
    bt = formula.formula_tensor("micro/convt.b", (2,), 1.0)
    xc = formula.formula_tensor("micro/convt.x", (1, 3, 2, 3), 1.0)
    np.testing.assert_allclose(np_ops.convT2x2s2(xc, wt, bt), g["micro_convt_y"], rtol=1e-6, atol=1e-6)


def _loss_inputs():
    cov_u8 = formula.synthetic_images(4, 32, 32, seed=31)
    st_u8 = cov_u8.copy()
    st_u8[0] = formula.lsbr_embed(cov_u8[0], 0.4, seed=1)
    st_u8[2] = formula.lsbr_embed(cov_u8[2], 1.0, seed=2)
    covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None]
    inputs = torch.from_numpy(st_u8.astype(np.float32) / np.float32(255.))[:, None]
    alphas = torch.tensor([0.4, 0.0, 1.0, 0.0])
    return covers, inputs, alphas


def test_losses_and_meters(golden):
    g = golden["losses"]
    covers, inputs, alphas = _loss_inputs()
    outputs = torch.from_numpy(g["loss_outputs"]).requires_grad_(True)
    for i, name in enumerate(["l1", "l2", "ws", "l1ws"]):
        outputs.grad = None
        v = losses_ref.LOSSES[name](outputs, (covers, alphas), inputs)
        v.backward()
        assert math.isclose(v.item(), float(g["loss_values"][i]), rel_tol=1e-6), name
        np.testing.assert_allclose(outputs.grad.numpy(), g[f"loss_{name}_dout"], rtol=1e-5, atol=1e-10)
    # numpy restatement of L1WS forward + analytic gradient
    lv, grad = np_ops.l1ws_loss(g["loss_outputs"], covers.numpy(), alphas.numpy(), inputs.numpy())
    assert math.isclose(lv, float(g["loss_values"][3]), rel_tol=1e-5)
    np.testing.assert_allclose(grad, g["loss_l1ws_dout"], rtol=1e-4, atol=1e-9)
    mae, ws = metrics_ref.RunningMean(), metrics_ref.RunningMean()
    o = g["loss_outputs"]
    for s in (slice(0, 2), slice(2, 4)):
        mae.push(metrics_ref.mae_batch_value(covers.numpy()[s], o[s]))
        ws.push(metrics_ref.ws_batch_value(inputs.numpy()[s], o[s], alphas.numpy()[s]))
    np.testing.assert_allclose([mae.avg, ws.avg], g["meter_values"], rtol=1e-6)


def test_adamw_numpy_matches_torch_optimizer(golden):
    g = golden["adamw"]
    sd = formula.formula_state_dict(0, "he")
    m = unet_ref.build_ref(0, sd)
    cov_u8 = formula.synthetic_images(2, 32, 32, seed=41)
    st_u8 = cov_u8.copy(); st_u8[1] = formula.lsbr_embed(cov_u8[1], 0.4, seed=9)
    covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None]
    inputs = torch.from_numpy(st_u8.astype(np.float32) / np.float32(255.))[:, None]
    alphas = torch.tensor([0.0, 0.4])
    state = {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in sd.items()}
    losses = []
    for step in (1, 2, 3):
        m.zero_grad()
        loss = losses_ref.l1ws_loss(m(inputs.clone()), (covers, alphas), inputs)
        loss.backward()
        losses.append(loss.item())
        with torch.no_grad():
            for k, p in m.named_parameters():
                pn, mn, vn = np_ops.adamw_step(p.numpy(), p.grad.numpy().astype(np.float64), *state[k], step)
                state[k] = (mn, vn)
                p.copy_(torch.from_numpy(pn))
    np.testing.assert_allclose(losses, g["adamw_losses"], rtol=1e-5)
    for k, p in m.named_parameters():
        np.testing.assert_allclose(p.detach().numpy(), g[f"adamw_{k}"], rtol=0, atol=2e-7, err_msg=k)


def test_ws_stats_and_png_kat():
    """Data-plumbing known answer from the reference's results/prediction/filters.csv
    (KB / AVG 3x3 predictor MAE on images/10.png) + predict_unet statistics semantics."""
    from PIL import Image
    kat = json.loads((GOLDEN / "filters_kat.json").read_text())
    x = np.array(Image.open(GOLDEN / "cover_10.png"))
    assert x.shape == (512, 512) and x.dtype == np.uint8
    xf = x.astype(np.float64)
    kb = (2 * (xf[:-2, 1:-1] + xf[2:, 1:-1] + xf[1:-1, :-2] + xf[1:-1, 2:])
          - (xf[:-2, :-2] + xf[:-2, 2:] + xf[2:, :-2] + xf[2:, 2:])) / 4.
    avg = (xf[:-2, 1:-1] + xf[2:, 1:-1] + xf[1:-1, :-2] + xf[1:-1, 2:]
           + xf[:-2, :-2] + xf[:-2, 2:] + xf[2:, :-2] + xf[2:, 2:]) / 8.
    inner = xf[1:-1, 1:-1]
    assert math.isclose(np.mean(np.abs(inner - kb)), kat["images/10.png"]["mae_3_KB"], rel_tol=1e-12)
    assert math.isclose(np.mean(np.abs(inner - avg)), kat["images/10.png"]["mae_3_AVG"], rel_tol=1e-12)
    # WS statistic: with x_hat == x the residual is zero; with x_hat == x_bar beta_hat == 1
    beta, l1 = np_ops.ws_stats(x[1:-1, 1:-1], x[1:-1, 1:-1].astype(np.float32))
    assert beta == 0.0 and l1 == 0.0
    beta, l1 = np_ops.ws_stats(x[1:-1, 1:-1], (x[1:-1, 1:-1] ^ 1).astype(np.float32))
    assert beta == 1.0 and l1 == 1.0


# ---- WS payload estimator: oracle vs the reference's own attack() ------------------------------

def _ws_planes64():
    cov = formula.synthetic_images(3, 64, 64, seed=51)
    return [cov[0], formula.lsbr_embed(cov[1], 0.4, seed=3), formula.lsbr_embed(cov[2], 1.0, seed=4)]


def _cross_mean(x):
    return ((x[:-2, 1:-1] + x[2:, 1:-1] + x[1:-1, :-2] + x[1:-1, 2:]) * np.float32(0.25))[..., :1]


def test_ws_attack_oracle_vs_reference(golden):
    """The reference evaluates its 3x3 convolutions through scipy's float32 FFT branch, so agreement is to FFT round-off
    (rel 2e-4 / abs 2e-5), not bitwise; see oracle/ws_ref.py."""
    from oracle import ws_ref
    from ws_unet_amd import filters
    g = golden["ws_attack"]
    for k in ("KB", "AVG", "AVG9", "1"):
        np.testing.assert_array_equal(filters.NAMED_FILTERS_2D[k], g[f"named2d_{k}"])
        np.testing.assert_array_equal(filters.NAMED_FILTERS_2D[k], g[f"named_{k}"])
    planes = _ws_planes64()
    np.testing.assert_allclose(ws_ref.filter_infere_single(planes[1].astype(np.float32)[..., None], filters.NAMED_FILTERS_2D["KB"])[..., 0],
                               g["filter64_KB"], atol=2e-4)
    ests = {"KB": lambda x: ws_ref.filter_infere_single(x, filters.NAMED_FILTERS_2D["KB"]),
            "AVG": lambda x: ws_ref.filter_infere_single(x, filters.NAMED_FILTERS_2D["AVG"]),
            "cross": _cross_mean}
    for name, est in ests.items():
        for i, p in enumerate(planes):
            for j, (w, cb) in enumerate(g["cfgs"]):
                got = ws_ref.attack_array(p, est, filters.NAMED_FILTERS_2D["AVG"], correct_bias=bool(cb), weighted=int(w))
                assert math.isclose(got, g[f"beta64_{name}"][i, j], rel_tol=2e-4, abs_tol=2e-5), (name, i, w, cb, got)
    for i, p in enumerate(planes):
        got = ws_ref.attack_array(p, ests["KB"], filters.NAMED_FILTERS_2D["AVG9"], weighted=1)
        assert math.isclose(got, g["beta64_KB_meanAVG9"][i], rel_tol=2e-4, abs_tol=2e-5)
