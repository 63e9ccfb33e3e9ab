"""GPU tests of the predictor / evaluate API against the oracle restatement of src/unet/evaluate.py."""
import json

import numpy as np
import pandas as pd
import pytest
import torch
from PIL import Image

from conftest import GOLDEN
from gpu_util import DEV, gpu_model, DEFAULT_MODE
from ws_unet_amd import evaluate, formula, get_unet_estimator
from ws_unet_amd.imread import imread4_f32
from oracle import evaluate_ref, unet_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref_model():
    return unet_ref.build_ref(2, formula.formula_state_dict(2, "he"))


# (mode, max |prediction - oracle| in 0..255 units).  None = what get_pretrained(mode=None) builds: 'f16f4p' (measured: max 0.07, mean 0.006)
@pytest.mark.parametrize("mode,tol", [("f32", 3e-3), ("bf16x3", 1e-2), ("f16f8", 3e-2), ("f16f8p", 3e-2), ("f16f4p", 1.5e-1), (None, 1.5e-1)])
def test_infere_single_and_predict_unet_on_real_cover(ref_model, mode, tol):
    """cover_10.png is one of the reference's own 512x512 covers; tolerances are in 0..255 units
    (tol/255 on the [0,1] output: 1.2e-5 / 4e-5 / 1.2e-4 / 6e-4, max over 260k pixels).  mode None is the DEFAULT a user of get_pretrained /
    get_unet_estimator gets (planar storage, fp4 cross terms: 'f16f4p'; drop_rate 0.)."""
    fname = GOLDEN / "cover_10.png"
    x = imread4_f32(fname)[..., 3:]
    model = gpu_model(2, "he", mode, drop_rate=0.)                 # get_pretrained builds with drop_rate=0.
    y = evaluate.infere_single(x, model)
    y_ref = evaluate_ref.infere_single(x, ref_model)
    assert y.shape == (510, 510, 1) and y.dtype == np.float32
    if mode is None:
        assert model.mode == DEFAULT_MODE and model.input_dropout is not None
    assert np.abs(y - y_ref).max() <= tol
    assert np.abs(y - y_ref).mean() <= tol / 10                    # 'f16f8p': MAE 4e-6 on the [0,1] output = 1e-3 in these units; 'f16f4p': 2.5e-5 = 6e-3
    assert x.max() > 1.0                                            # the caller's array is not modified in place
    res = evaluate.predict_unet(fname, model, name="images/10.png", height=512, width=512)
    ref = evaluate_ref.predict_unet_array(x, ref_model)
    assert set(res) == {"name", "height", "width", "beta_hat", "l1"}
    stat_tol = 1e-3 if mode in (None, "f16f4p") else 1e-4           # beta_hat / l1 are means over the image: they follow the MAE
    assert abs(res["beta_hat"] - ref["beta_hat"]) <= stat_tol and abs(res["l1"] - ref["l1"]) <= stat_tol


def _make_dataset(root, n=5):
    (root / "images").mkdir()
    u8 = formula.synthetic_images(n, 512, 512, seed=50)
    names = []
    for i, k in enumerate((3, 10, 1, 22, 7)[:n]):
        Image.fromarray(u8[i]).save(root / "images" / f"{k}.png")
        names.append(f"images/{k}.png")
    (root / "images" / "files.csv").write_text("name,height,width\n" + "".join(f"{n_},512,512\n" for n_ in names))
    sdir = root / "stego_LSBR_alpha_0.4"
    sdir.mkdir()
    for i, k in enumerate((3, 10)):
        Image.fromarray(formula.lsbr_embed(u8[i], 0.4, seed=k)).save(sdir / f"{k}.png")
    (sdir / "files.csv").write_text("name,height,width,stego_method,alpha\n" + "".join(
        f"stego_LSBR_alpha_0.4/{k}.png,512,512,LSBR,0.4\n" for k in (3, 10)))
    return u8


@pytest.mark.parametrize("mode", ["f32", None])
def test_evaluate_loop_per_image_vs_batched(tmp_path, ref_model, mode):
    u8 = _make_dataset(tmp_path)
    model = gpu_model(2, "he", mode, drop_rate=0.)                  # None: the default planar mode, as get_pretrained builds it
    df = evaluate.predict_unet_cover(tmp_path, model=model, progress_on=False)
    dfb = evaluate.predict_unet_cover_batched(tmp_path, model=model)
    assert list(df.columns) == ["name", "height", "width", "beta_hat", "l1"] == list(dfb.columns)
    assert df["name"].tolist() == ["images/1.png", "images/10.png", "images/22.png", "images/3.png", "images/7.png"] == dfb["name"].tolist()
    np.testing.assert_allclose(dfb["beta_hat"].to_numpy(float), df["beta_hat"].to_numpy(float), atol=2e-5)
    np.testing.assert_allclose(dfb["l1"].to_numpy(float), df["l1"].to_numpy(float), atol=2e-5)
    # oracle for the first row (images/1.png is u8[2])
    ref = evaluate_ref.predict_unet_array(u8[2][..., None].astype(np.float32), ref_model)
    # l1 is a mean |x - prediction| in 0..255 units: it follows the mode's MAE (default 'f16f4p': 2.5e-5 x 255 = 6e-3; measured difference 8e-4)
    assert abs(df["beta_hat"][0] - ref["beta_hat"]) <= 1e-4 and abs(df["l1"][0] - ref["l1"]) <= (5e-3 if mode is None else 1e-4)
    st = evaluate.predict_unet_stego(tmp_path, model=model, stego_method="LSBR")
    stb = evaluate.predict_unet_stego_batched(tmp_path, model=model, stego_method="LSBR", alpha=0.4)
    assert list(st.columns) == ["name", "height", "width", "stego_method", "alpha", "beta_hat", "l1"] == list(stb.columns)
    np.testing.assert_allclose(stb["beta_hat"].to_numpy(float), st["beta_hat"].to_numpy(float), atol=2e-5)
    take = evaluate.predict_unet_cover_batched(tmp_path, model=model, take_num_images=2, skip_num_images=1)
    assert take["name"].tolist() == ["images/10.png", "images/22.png"]


def test_model_discovery_and_checkpoint_roundtrip(tmp_path):
    sd = formula.formula_state_dict(2, "he")
    run = tmp_path / "LSBR" / "240101000000-1-unet_2-test"
    (run / "model").mkdir(parents=True)
    cfg = {"stego_method": "LSBR", "alpha": "0.400", "loss": "l1ws", "network": "unet_2", "drop_rate": 0.0, "debug": False}
    (run / "config.json").write_text(json.dumps(cfg))
    torch.save({"epoch": 7, "state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}, "best_val_loss": 0.1,
                "patience": 3, "optimizer": None, "scheduler": None}, run / "model" / "best_model.pt.tar")
    dbg = tmp_path / "LSBR" / "debug-run"
    (dbg / "model").mkdir(parents=True)
    (dbg / "config.json").write_text(json.dumps({**cfg, "debug": True}))
    torch.save({"epoch": 1, "state_dict": {}}, dbg / "model" / "best_model.pt.tar")
    nock = tmp_path / "LSBR" / "no-checkpoint"
    nock.mkdir()
    (nock / "config.json").write_text(json.dumps(cfg))
    name = evaluate.get_model_name("LSBR", model_dir=tmp_path)
    assert name == run.name
    assert evaluate.get_model_config(tmp_path, "LSBR", name)["network"] == "unet_2"
    model = evaluate.get_pretrained(tmp_path / "LSBR", (3,), model_name=name, mode="f32")
    assert model.nsteps == 2 and model.input_dropout is not None and next(model.parameters()).is_cuda
    np.testing.assert_array_equal(model.e32.weight.detach().cpu().numpy(), sd["e32.weight"])
    with pytest.raises(RuntimeError, match="no model for"):
        evaluate.get_model_name("HILLR", model_dir=tmp_path)
    dup = tmp_path / "LSBR" / "second-run"
    (dup / "model").mkdir(parents=True)
    (dup / "config.json").write_text(json.dumps(cfg))
    torch.save({"epoch": 2, "state_dict": {}}, dup / "model" / "best_model.pt.tar")
    with pytest.raises(RuntimeError, match="multiple models for"):
        evaluate.get_model_name("LSBR", model_dir=tmp_path)
    # the closure used by the WS estimator callers
    predict = get_unet_estimator(tmp_path / "LSBR", (3,), model_name=name)
    x = formula.synthetic_images(1, 512, 512, seed=3)[0][..., None].astype(np.float32)
    y = predict(x)
    assert y.shape == (510, 510, 1) and 0 <= y.min() and y.max() <= 255
    # the DEFAULT closure ('f16f4p', drop_rate 0.) against the oracle's infere_single on the same weights (0..255 units)
    y_ref = evaluate_ref.infere_single(x, unet_ref.build_ref(2, sd))
    assert np.abs(y - y_ref).max() <= 1.5e-1 and np.abs(y - y_ref).mean() <= 1.5e-2
    from ws_unet_amd.model import get_model
    with pytest.raises(NotImplementedError):
        get_model("cnn_1", in_channels=1)


@pytest.mark.parametrize("mode", ["f32", None])
def test_sharded_dataset_evaluate_and_cli(tmp_path, mode):
    """predict_unet_sharded == the per-image iterators' table (single rank), and the `python -m ws_unet_amd.evaluate` driver writes it
    (mode None: the driver's and get_pretrained's default, 'f16f4p')."""
    data = tmp_path / "data"
    data.mkdir()
    _make_dataset(data)
    model = gpu_model(2, "he", mode, drop_rate=0.)
    ref_c = evaluate.predict_unet_cover(data, model=model, progress_on=False)
    got_c = evaluate.predict_unet_sharded(data, model, batch_size=2)
    assert got_c["name"].tolist() == ref_c["name"].tolist() and list(got_c.columns) == list(ref_c.columns)
    np.testing.assert_allclose(got_c["beta_hat"].to_numpy(float), ref_c["beta_hat"].to_numpy(float), atol=2e-5)
    np.testing.assert_allclose(got_c["l1"].to_numpy(float), ref_c["l1"].to_numpy(float), atol=2e-5)
    ref_s = evaluate.predict_unet_stego(data, model=model, stego_method="LSBR")
    got_s = evaluate.predict_unet_sharded(data, model, stego_method="LSBR")
    assert got_s["name"].tolist() == ref_s["name"].tolist() and set(ref_s.columns) <= set(got_s.columns)
    np.testing.assert_allclose(got_s["beta_hat"].to_numpy(float), ref_s["beta_hat"].to_numpy(float), atol=2e-5)
    # driver: a model directory with one run, covers + LSBR rows -> CSV
    sd = formula.formula_state_dict(2, "he")
    run = tmp_path / "models" / "LSBR" / "run-x"
    (run / "model").mkdir(parents=True)
    (run / "config.json").write_text(json.dumps({"stego_method": "LSBR", "alpha": "0.400", "loss": "l1ws", "network": "unet_2",
                                                 "drop_rate": 0.0, "debug": False}))
    torch.save({"epoch": 1, "state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}}, run / "model" / "best_model.pt.tar")
    out = tmp_path / "res" / "ws.csv"
    evaluate.main(["--data", str(data), "--model-dir", str(tmp_path / "models"), "--stego-method", "LSBR", "--eval-methods", "LSBR",
                   "--out", str(out)] + (["--mode", mode] if mode else []))
    table = pd.read_csv(out)
    assert len(table) == 7 and table["name"].tolist()[:5] == ref_c["name"].tolist()
    np.testing.assert_allclose(table["beta_hat"].to_numpy(float)[:5], ref_c["beta_hat"].to_numpy(float), atol=2e-5)
    # the same driver fed from pre-decoded uint8 shards (written on first use): the identical CSV
    out2 = tmp_path / "res" / "ws_shards.csv"
    try:
        evaluate.main(["--data", str(data), "--model-dir", str(tmp_path / "models"), "--stego-method", "LSBR", "--eval-methods", "LSBR",
                       "--out", str(out2), "--u8-shards", str(tmp_path / "shards")] + (["--mode", mode] if mode else []))
    finally:
        evaluate.use_u8_shards(None)
    assert (tmp_path / "shards" / "index.json").exists()
    pd.testing.assert_frame_equal(pd.read_csv(out2), table)


def test_range_guard_looks_once_per_dataset_pass(tmp_path, caplog):
    """The +-448 guard of the default mode is read once per data-set pass by the batched / sharded drivers: a model
    that HAS run a clean forward (so the first-forward check is spent) gets new weights through `param.data`-style edits the version
    counter does not see as a new checkpoint; the next pass must notice, warn, and return the 'bf16x3s' table."""
    import logging
    _make_dataset(tmp_path)
    clean = gpu_model(2, "he", None, drop_rate=0.)
    ref = evaluate.predict_unet_cover_batched(tmp_path, model=clean)
    assert clean.mode == DEFAULT_MODE
    for driver in ("batched", "sharded", "per_image"):
        m = gpu_model(2, "he", None, drop_rate=0.)
        x0 = torch.zeros(1, 1, 16, 16, device=DEV)
        with torch.no_grad():
            m(x0)                                                     # first forward: range check spent on harmless activations
        assert m.mode == DEFAULT_MODE and m._range_checked
        with torch.no_grad():
            m.e11.weight.data.mul_(3000.0); m.e11.bias.data.mul_(3000.0); m.e12.weight.data.div_(3000.0)
        m._pack_cache.clear()                                         # packed weights follow, the range check stays spent
        assert m._range_checked
        caplog.clear()
        with caplog.at_level(logging.WARNING):
            if driver == "batched":
                got = evaluate.predict_unet_cover_batched(tmp_path, model=m)
            elif driver == "sharded":
                got = evaluate.predict_unet_sharded(tmp_path, m, batch_size=2)
            else:
                got = evaluate.predict_unet_cover(tmp_path, model=m, progress_on=False)
        assert m.mode == "bf16x3s", driver
        assert any("beyond" in r.message for r in caplog.records), driver
        assert got["name"].tolist() == ref["name"].tolist()
        # same network function (scaling e11 up and e12 down by the same factor), now computed in fp32-range storage
        np.testing.assert_allclose(got["beta_hat"].to_numpy(float), ref["beta_hat"].to_numpy(float), atol=2e-3)
        np.testing.assert_allclose(got["l1"].to_numpy(float), ref["l1"].to_numpy(float), atol=2e-3)


def test_load_state_dict_rearms_the_range_check(caplog):
    """a model that already ran a forward and then receives new weights must be range-checked again (load_state_dict and
    invalidate_packed re-arm the one-time look); the optimiser's per-step invalidation must not (the trainer polls per epoch)."""
    import logging
    m = gpu_model(1, "he", None)
    x = torch.rand((1, 1, 32, 64), generator=torch.Generator().manual_seed(2)).to(DEV)
    with torch.no_grad():
        m(x)
    assert m._range_checked and m.mode == DEFAULT_MODE
    m.invalidate_packed(recheck_range=False)
    assert m._range_checked
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    sd["e11.weight"] *= 3000.0; sd["e11.bias"] *= 3000.0; sd["e12.weight"] /= 3000.0
    m.load_state_dict(sd)
    assert not m._range_checked and not m._range_checked_train
    with caplog.at_level(logging.WARNING), torch.no_grad():
        y = m(x)
    assert m.mode == "bf16x3s" and any("beyond" in r.message for r in caplog.records)
    ref = unet_ref.unet_forward(x.cpu().clone(), {k: v.cpu() for k, v in sd.items()}, 1)
    assert (y.cpu() - ref).abs().max().item() <= 1e-4


def test_u8_shards_give_the_identical_table(tmp_path):
    """VERDICT r03 next #7d: the batched evaluate pass fed from pre-decoded uint8 shards (evaluate.write_u8_shards / use_u8_shards) returns the
    table of the PNG-fed pass, value for value (the planes are the same bytes; src/fabrika.py:85-89 row order)."""
    _make_dataset(tmp_path)
    model = gpu_model(2, "he", None, drop_rate=0.)
    df0 = evaluate.predict_unet_cover_batched(tmp_path, model=model)
    st0 = evaluate.predict_unet_stego_batched(tmp_path, model=model, stego_method="LSBR")
    files = [str(tmp_path / n) for n in df0["name"].tolist() + st0["name"].tolist()]
    try:
        assert evaluate.use_u8_shards(evaluate.write_u8_shards(files, tmp_path / "shards", images_per_shard=4)) == len(files)
        df1 = evaluate.predict_unet_cover_batched(tmp_path, model=model)
        st1 = evaluate.predict_unet_stego_batched(tmp_path, model=model, stego_method="LSBR")
    finally:
        evaluate.use_u8_shards(None)
    for a, b in ((df0, df1), (st0, st1)):
        assert a["name"].tolist() == b["name"].tolist()
        np.testing.assert_array_equal(a["beta_hat"].to_numpy(), b["beta_hat"].to_numpy())
        np.testing.assert_array_equal(a["l1"].to_numpy(), b["l1"].to_numpy())


def test_per_image_api_one_readback(tmp_path, caplog):
    """Round 4: predict_unet brings beta_hat, l1 AND the range flag back in one copy.  Same numbers as the batched path;
    a tripped range flag still switches the model (loudly) and the row is recomputed in the fallback arithmetic."""
    _make_dataset(tmp_path)
    model = gpu_model(2, "he", None, drop_rate=0.)
    per = evaluate.predict_unet_cover(tmp_path, model=model, progress_on=False)
    bat = evaluate.predict_unet_cover_batched(tmp_path, model=model)
    np.testing.assert_allclose(per["beta_hat"].to_numpy(float), bat["beta_hat"].to_numpy(float), atol=2e-5)
    x = evaluate.load_planes_u8([str(tmp_path / "images" / "1.png")]).to(DEV)
    b0, l0, tripped = evaluate.predict_u8_one_readback(x, model)
    assert not tripped and b0.shape == (1,) and np.float32(b0[0]) == np.float32(per["beta_hat"].iloc[0])
    model._range_flag_tensor(torch.device(DEV)).fill_(1)                   # as if an epilogue had stored a value beyond +-448
    import logging
    with caplog.at_level(logging.WARNING):
        row = evaluate.predict_unet(str(tmp_path / "images" / "1.png"), model)
    assert model.mode == "bf16x3s" and any("448" in r.message for r in caplog.records)
    assert abs(float(row["beta_hat"]) - float(b0[0])) < 1e-4


def _many_pngs(root, n, seed=77):
    (root / "images").mkdir()
    u8 = formula.synthetic_images(n, 512, 512, seed=seed)
    for i in range(n):
        Image.fromarray(u8[i]).save(root / "images" / f"{i:03d}.png", compress_level=1)
    (root / "images" / "files.csv").write_text("name,height,width\n" + "".join(f"images/{i:03d}.png,512,512\n" for i in range(n)))
    return u8


def test_per_image_api_rows_ahead_ride_along(tmp_path, monkeypatch):
    """The per-image API (the reference's call pattern, evaluate.py:48,142-149): rows announced ahead whose files are already decoded join the
    launch of the row that is asked for (at most _MICRO_BATCH images), their statistics wait for their own calls.  Rows, order and numbers are
    those of the one-image-per-launch loop, bit for bit; every image is computed exactly once."""
    n = 40
    _many_pngs(tmp_path, n)
    model = gpu_model(2, "he", None, drop_rate=0.)
    sizes = []
    real = evaluate.predict_u8_batch
    monkeypatch.setattr(evaluate, "predict_u8_batch", lambda x, m: (sizes.append(int(x.shape[0])), real(x, m))[1])
    df = evaluate.predict_unet_cover(tmp_path, model=model, progress_on=False)
    assert sum(sizes) == n and max(sizes) > 1 and max(sizes) <= evaluate._MICRO_BATCH, sizes
    assert not evaluate._AHEAD["results"] and not evaluate._AHEAD["pending"] and not evaluate._AHEAD["inflight"]          # nothing left behind
    sizes.clear()
    monkeypatch.setattr(evaluate, "_MICRO_BATCH", 1)
    df1 = evaluate.predict_unet_cover(tmp_path, model=model, progress_on=False)
    assert sizes == [1] * n
    assert df["name"].tolist() == df1["name"].tolist() == [f"images/{i:03d}.png" for i in range(n)]
    np.testing.assert_array_equal(df["beta_hat"].to_numpy(), df1["beta_hat"].to_numpy())
    np.testing.assert_array_equal(df["l1"].to_numpy(), df1["l1"].to_numpy())
    assert df["beta_hat"].dtype == df1["beta_hat"].dtype


def test_rows_computed_ahead_follow_the_file_and_the_model(tmp_path):
    """A result computed ahead is used only for the same file contents, the same model and the same arithmetic; a pass that ends (or raises)
    leaves none behind."""
    u8 = _many_pngs(tmp_path, 6, seed=5)
    files = [str(tmp_path / "images" / f"{i:03d}.png") for i in range(6)]
    model = gpu_model(2, "he", None, drop_rate=0.)
    plain = [evaluate.predict_unet(f, model) for f in files]                        # nothing announced: one image per launch
    evaluate._lookahead_reset()
    for f in files[1:]:
        evaluate._lookahead(f)
    for fut, _ in list(evaluate._AHEAD["pending"].values()):
        fut.result()                                                                # all five decoded
    r0 = evaluate.predict_unet(files[0], model)                                      # ... and computed with row 0
    assert set(evaluate._AHEAD["results"]) == set(files[1:]) and not evaluate._AHEAD["inflight"] and r0["beta_hat"] == plain[0]["beta_hat"]
    # file 1 is rewritten before its row comes: its ahead result is dropped, the row is computed from the new file
    Image.fromarray(u8[5]).save(files[1], compress_level=1)
    r1 = evaluate.predict_unet(files[1], model)
    assert r1["beta_hat"] == plain[5]["beta_hat"] and r1["l1"] == plain[5]["l1"]
    # another model object: not its result
    other = gpu_model(2, "default", None, drop_rate=0.)
    r2 = evaluate.predict_unet(files[2], other)
    assert r2["beta_hat"] != plain[2]["beta_hat"]
    # the same model: taken from the cache, same numbers as the plain loop
    r3 = evaluate.predict_unet(files[3], model)
    assert r3["beta_hat"] == plain[3]["beta_hat"] and r3["l1"] == plain[3]["l1"] and files[3] not in evaluate._AHEAD["results"]
    evaluate._lookahead_reset()
    assert not evaluate._AHEAD["results"]
