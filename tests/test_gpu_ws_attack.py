"""GPU tests of the WS payload estimator (reference src/ws/estimate.py) through the C ABI.

Checkers: oracle/ws_ref.py (direct fp32 evaluation) at rel 2e-6, and tests/golden/ws_attack.npz (the reference's own
`attack`, whose scipy FFT convolutions carry float32 round-off) at rel 2e-4 / abs 2e-5.
"""
import json
import math

import numpy as np
import pytest
import torch
from PIL import Image

from gpu_util import DEV, gpu_model
from ws_unet_amd import filters, formula, ops
from ws_unet_amd.imread import imread4_u8
from ws_unet_amd.ws import estimate
from oracle import ws_ref

pytestmark = pytest.mark.gpu

AVG = filters.NAMED_FILTERS_2D["AVG"]


def _planes64():
    cov = formula.synthetic_images(3, 64, 64, seed=51)
    return [cov[0], formula.lsbr_embed(cov[1], 0.4, seed=3), formula.lsbr_embed(cov[2], 1.0, seed=4)]


def cross_mean(x):
    return ((x[:-2, 1:-1] + x[2:, 1:-1] + x[1:-1, :-2] + x[1:-1, 2:]) * np.float32(0.25))[..., :1]


def _close(a, b, rel, abs_):
    return math.isclose(float(a), float(b), rel_tol=rel, abs_tol=abs_)


@pytest.mark.parametrize("name", ["KB", "AVG", "cross"])
def test_statistic_vs_oracle_and_reference_golden(golden, name):
    g = golden["ws_attack"]
    planes = _planes64()
    est = cross_mean if name == "cross" else filters.get_filter_estimator(filter_name=name, flatten=False)
    oracle_est = cross_mean if name == "cross" else (lambda x: ws_ref.filter_infere_single(x, filters.NAMED_FILTERS_2D[name]))
    x_u8 = torch.from_numpy(np.stack(planes)).to(DEV)
    for j, (w, cb) in enumerate(g["cfgs"]):
        beta = estimate._stat(x_u8, est, AVG, int(w), bool(cb),
                              host_planes=[p.astype(np.float32)[..., None] for p in planes]).cpu().numpy()
        for i, p in enumerate(planes):
            ref = ws_ref.attack_array(p, oracle_est, AVG, correct_bias=bool(cb), weighted=int(w))
            assert _close(beta[i], ref, 2e-6, 2e-7), (name, i, w, cb, beta[i], ref)
            assert _close(beta[i], g[f"beta64_{name}"][i, j], 2e-4, 2e-5), (name, i, w, cb)


def test_general_mean_filter_and_sums(golden):
    g = golden["ws_attack"]
    planes = _planes64()
    x_u8 = torch.from_numpy(np.stack(planes)).to(DEV)
    kb = filters.NAMED_FILTERS_2D["KB"]
    beta, sums = ops.ws_attack(x_u8, None, pixel_filter=kb, mean_filter=filters.NAMED_FILTERS_2D["AVG9"], weighted=1, return_sums=True)
    for i in range(3):
        assert _close(beta[i].item(), g["beta64_KB_meanAVG9"][i], 2e-4, 2e-5)
    s = sums.cpu().numpy()
    np.testing.assert_allclose(np.maximum(s[:, 1] / s[:, 0], 0), beta.cpu().numpy(), rtol=1e-6)
    # uniform weights: sum w = number of interior pixels exactly
    _, s0 = ops.ws_attack(x_u8, None, pixel_filter=kb, weighted=0, return_sums=True)
    np.testing.assert_array_equal(s0[:, 0].cpu().numpy(), np.full(3, 62 * 62, np.float64))


def test_filter_predictor_kernel(golden):
    g = golden["ws_attack"]
    p = _planes64()[1].astype(np.float32)[..., None]
    y = filters.infere_single(p, filters.NAMED_FILTERS_2D["KB"])
    assert y.shape == (62, 62, 1) and y.dtype == np.float32
    np.testing.assert_array_equal(y, ws_ref.filter_infere_single(p, filters.NAMED_FILTERS_2D["KB"]))   # same tap order, same roundings
    np.testing.assert_allclose(y[..., 0], g["filter64_KB"], atol=2e-4)
    with pytest.raises(NotImplementedError):
        filters.infere_single(p, np.zeros((3, 3, 2), np.float32))


def test_lsb_delta_and_determinism_and_ragged():
    u = torch.arange(256, dtype=torch.uint8, device=DEV)
    d = ops.lsb_delta_unit(u).cpu().numpy()
    un = np.arange(256, dtype=np.uint8)
    np.testing.assert_array_equal(d, ((un ^ 1).astype(np.float32) - un.astype(np.float32)) / np.float32(255.))
    for (n, h, w) in [(1, 3, 3), (2, 5, 300), (3, 130, 7), (33, 40, 40)]:
        x = formula.synthetic_images(n, h, w, seed=60 + h)
        x_u8 = torch.from_numpy(x).to(DEV)
        hat = torch.from_numpy(formula.formula_tensor(f"ws/hat{h}", (n, h - 2, w - 2), 255.0)).abs().to(DEV)
        b1 = ops.ws_attack(x_u8, hat, mean_filter=AVG, hat_scale=1.0, weighted=1)
        b2 = ops.ws_attack(x_u8, hat, mean_filter=AVG, hat_scale=1.0, weighted=1)
        assert torch.equal(b1, b2)
        for i in range(n):
            ref = ws_ref.attack_array(x[i], lambda _x, i=i: hat[i].cpu().numpy()[..., None], AVG, weighted=1)
            assert _close(b1[i].item(), ref, 2e-6, 2e-7), (n, h, w, i)
        # the full-frame layout reads the same interior
        full = torch.zeros((n, h, w), device=DEV)
        full[:, 1:-1, 1:-1] = hat
        assert torch.equal(ops.ws_attack(x_u8, full, mean_filter=AVG, hat_scale=1.0, weighted=1), b1)


def test_argument_errors():
    x_u8 = torch.zeros((1, 8, 8), dtype=torch.uint8, device=DEV)
    hat = torch.zeros((1, 6, 6), device=DEV)
    with pytest.raises(Exception, match="exactly one"):
        ops.ws_attack(x_u8, hat, pixel_filter=AVG, mean_filter=AVG)
    with pytest.raises(Exception, match="mean_filter"):
        ops.ws_attack(x_u8, hat, weighted=1)
    with pytest.raises(Exception, match="correct_bias"):
        ops.ws_attack(x_u8, hat, mean_filter=AVG, correct_bias=True)
    with pytest.raises(ValueError):
        ops.ws_attack(x_u8, torch.zeros((1, 5, 5), device=DEV), mean_filter=AVG)
    with pytest.raises(Exception, match="CPU tensor"):
        ops.ws_attack(x_u8.cpu(), hat, mean_filter=AVG)


@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("bf16x3", 1e-4)])
def test_unet_estimator_vs_reference_golden(golden, mode, tol):
    """512x512, formula 'he' weights: reference attack() with the reference UNet as pixel_estimator (beta512_unet)."""
    g = golden["ws_attack"]
    c512 = formula.synthetic_images(1, 512, 512, seed=7)[0]
    planes = [c512, formula.lsbr_embed(c512, 0.4, seed=5)]
    est = estimate.UNetEstimator(gpu_model(2, "he", mode, drop_rate=0.))
    x_u8 = torch.from_numpy(np.stack(planes)).to(DEV)
    for j, (w, cb) in enumerate(g["cfg512"]):
        beta = estimate._stat(x_u8, est, AVG, int(w), bool(cb)).cpu().numpy()
        np.testing.assert_allclose(beta, g["beta512_unet"][:, j], rtol=2e-4, atol=tol, err_msg=f"{w} {cb}")
    with pytest.raises(ValueError, match="512x512"):
        estimate._stat(x_u8[:, :64, :64].contiguous(), est, AVG, 1, False)


def _make_dataset(root, size=64, n=4):
    (root / "images").mkdir()
    u8 = formula.synthetic_images(n, size, size, seed=70)
    keys = (3, 10, 1, 22)[:n]
    for i, k in enumerate(keys):
        Image.fromarray(u8[i]).save(root / "images" / f"{k}.png")
    (root / "images" / "files.csv").write_text("name,height,width\n" + "".join(f"images/{k}.png,{size},{size}\n" for k in keys))
    sdir = root / "stego_LSBR_alpha_0.4"
    sdir.mkdir()
    st = {}
    for i, k in enumerate(keys[:2]):
        st[k] = formula.lsbr_embed(u8[i], 0.4, seed=k)
        Image.fromarray(st[k]).save(sdir / f"{k}.png")
    (sdir / "files.csv").write_text("name,height,width,stego_method,alpha\n" + "".join(
        f"stego_LSBR_alpha_0.4/{k}.png,{size},{size},LSBR,0.4\n" for k in keys[:2]))
    return dict(zip(keys, u8)), st


def test_run_over_dataset_filters(tmp_path):
    """`run` (estimate.py:149-205) with a named linear predictor: per-image iterator == batched iterator == oracle."""
    covers, stegos = _make_dataset(tmp_path)
    res = estimate.run(tmp_path, None, None, "KB", None, (3,), correct_bias=False, weighted=1, progress_on=False)
    resb = estimate.run(tmp_path, None, None, "KB", None, (3,), correct_bias=False, weighted=1, batched=True, batch_size=3)
    assert res["name"].tolist() == ["images/1.png", "images/10.png", "images/22.png", "images/3.png"] == resb["name"].tolist()
    for col in ("beta_hat", "channels", "weighted", "correct_bias", "model_name"):
        assert col in res.columns and col in resb.columns
    assert res["channels"].tolist() == ["3"] * 4 and res["model_name"].tolist() == ["KB"] * 4
    np.testing.assert_array_equal(res["beta_hat"].to_numpy(np.float32), resb["beta_hat"].to_numpy(np.float32))
    kb = lambda x: ws_ref.filter_infere_single(x, filters.NAMED_FILTERS_2D["KB"])
    for name, b in zip(res["name"], res["beta_hat"]):
        k = int(name.split("/")[1].split(".")[0])
        assert _close(b, ws_ref.attack_array(covers[k], kb, AVG, weighted=1), 2e-6, 2e-7)
    st = estimate.run(tmp_path, "LSBR", 0.4, "AVG", None, (3,), correct_bias=True, weighted=0, batched=True)
    assert st["name"].tolist() == ["stego_LSBR_alpha_0.4/10.png", "stego_LSBR_alpha_0.4/3.png"]
    assert st["stego_method"].tolist() == ["LSBR"] * 2 and st["alpha"].tolist() == [0.4] * 2
    avg = lambda x: ws_ref.filter_infere_single(x, filters.NAMED_FILTERS_2D["AVG"])
    for name, b in zip(st["name"], st["beta_hat"]):
        k = int(name.split("/")[1].split(".")[0])
        assert _close(b, ws_ref.attack_array(stegos[k], avg, AVG, correct_bias=True, weighted=0), 2e-6, 2e-7)


def test_attack_result_layout_and_bad_estimator(tmp_path):
    covers, _ = _make_dataset(tmp_path, n=1)
    f = tmp_path / "images" / "3.png"
    proc = filters.get_processor_2d(channels=(3,))
    r = estimate.attack(f, (3,), cross_mean, imread=imread4_u8, process_image=proc, name="images/3.png", height=64)
    assert list(r) == ["name", "height", "beta_hat", "channels", "weighted", "correct_bias"]
    assert r["channels"] == "3" and r["weighted"] == 1 and r["correct_bias"] is False
    assert _close(r["beta_hat"], ws_ref.attack_array(covers[3], cross_mean, AVG), 2e-6, 2e-7)
    bad = estimate.attack(f, (3,), lambda x: x[2:-2, 2:-2, :1], imread=imread4_u8, process_image=proc)
    assert bad["beta_hat"] is None                                   # estimate.py:122-123
    df = estimate.attack_cover(tmp_path, channels=(3,), pixel_estimator=lambda x: x[2:-2, 2:-2, :1], imread=imread4_u8,
                               process_image=proc, progress_on=False)
    assert df["beta_hat"].isna().all()


def test_run_with_unet_checkpoint(tmp_path):
    """`run` with a model directory: get_unet_estimator -> UNetEstimator -> both network passes + statistic on the device."""
    (tmp_path / "data").mkdir()
    _make_dataset(tmp_path / "data", size=512, n=2)
    sd = formula.formula_state_dict(2, "he")
    rundir = tmp_path / "models" / "LSBR" / "run-a"
    (rundir / "model").mkdir(parents=True)
    (rundir / "config.json").write_text(json.dumps({"stego_method": "LSBR", "alpha": "0.400", "loss": "l1ws", "network": "unet_2",
                                                    "drop_rate": 0.0, "debug": False}))
    torch.save({"epoch": 1, "state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}}, rundir / "model" / "best_model.pt.tar")
    res = estimate.run(tmp_path / "data", None, None, "run-a", tmp_path / "models" / "LSBR", (3,), correct_bias=True, weighted=1,
                       progress_on=False)
    resb = estimate.run(tmp_path / "data", None, None, "run-a", tmp_path / "models" / "LSBR", (3,), correct_bias=True, weighted=1,
                        batched=True)
    assert res["model_name"].tolist() == ["UNet", "UNet"]
    np.testing.assert_allclose(res["beta_hat"].to_numpy(float), resb["beta_hat"].to_numpy(float), rtol=1e-5, atol=1e-6)
    # the generic host-callable route of `attack` gives the same number as the on-device route
    est = estimate.UNetEstimator(gpu_model(2, "he", None, drop_rate=0.))
    host = estimate.attack(tmp_path / "data" / "images" / "10.png", (3,), lambda x: est(x), correct_bias=True, weighted=1,
                           imread=imread4_u8, process_image=filters.get_processor_2d((3,)))
    row = res[res["name"] == "images/10.png"]["beta_hat"].iloc[0]
    assert _close(host["beta_hat"], row, 1e-5, 1e-6)


def test_estimate_driver_writes_the_result_table(tmp_path):
    """`python -m ws_unet_amd.ws.estimate` (reference estimate.py:208-275): filters + both UNets over covers and stego -> one CSV."""
    import pandas as pd
    (tmp_path / "data").mkdir()
    _make_dataset(tmp_path / "data", size=512, n=2)
    sd = formula.formula_state_dict(2, "he")
    for method, loss, dr in (("LSBR", "l1ws", 0.0), ("dropout", "l1", 0.1)):
        rundir = tmp_path / "models" / method / f"run-{method}"
        (rundir / "model").mkdir(parents=True)
        (rundir / "config.json").write_text(json.dumps({"stego_method": method, "alpha": "0.400" if method == "LSBR" else None, "loss": loss,
                                                        "network": "unet_2", "drop_rate": dr, "debug": False}))
        torch.save({"epoch": 1, "state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}}, rundir / "model" / "best_model.pt.tar")
    out = tmp_path / "res" / "ws.csv"
    estimate.main(["--data", str(tmp_path / "data"), "--model-dir", str(tmp_path / "models"), "--alphas", "0.4", "--out", str(out)])
    t = pd.read_csv(out)
    assert set(t["model_name"]) == {"AVG", "KB", "UNet_l1", "UNet_l1ws_LSBR"}
    assert set(t["stego_method"]) == {"Cover", "LSBR"}
    assert len(t) == 4 * (2 + 2)                                  # 4 predictors x (2 covers + 2 stego images)
    for col in ("name", "beta_hat", "channels", "weighted", "correct_bias"):
        assert col in t.columns
    assert (t["beta_hat"] >= 0).all() and t["weighted"].eq(0).all()
    # same checkpoint in both model directories -> both UNet variants give the same estimates
    a = t[t.model_name == "UNet_l1"].sort_values("name")["beta_hat"].to_numpy()
    b = t[t.model_name == "UNet_l1ws_LSBR"].sort_values("name")["beta_hat"].to_numpy()
    np.testing.assert_allclose(a, b, rtol=1e-6)
