"""Data-parallel train step on the GPU: an N-rank step on a sharded batch == the 1-rank step on the whole batch, replicas
are made identical by one flat broadcast, and early stopping is decided from all-reduced epoch sums so no rank leaves the loop alone.
Two fresh child processes share this box's one GPU and use the gloo backend on CUDA tensors (RCCL needs one GPU per rank); the code
path through Trainer / parallel is the one RCCL serves on a multi-GPU node (pattern: src/detector/train.py:55-95,281-304)."""
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from gpu_util import DEV, gpu_model, DEFAULT_MODE
from ws_unet_amd.trainer import Trainer

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent


def test_two_rank_step_equals_single_rank_step(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    outs = [tmp_path / f"rank{r}.npz" for r in range(2)]
    procs = [subprocess.Popen([sys.executable, str(HERE / "dp_worker.py"), str(r), "2", str(port), str(outs[r])],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:               # a rank that left the epoch loop alone strands the other in an all-reduce
            for q in procs:
                q.kill()
            pytest.fail("data-parallel ranks dead-locked (timeout)")
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    r0, r1 = np.load(outs[0]), np.load(outs[1])

    # single-process step on the concatenated batch
    sys.path.insert(0, str(HERE))
    from dp_worker import batch
    covers, inputs, alphas = batch()
    model = gpu_model(1, "he", "f32")
    tr = Trainer(model, loss="l1ws", lr=1e-3)
    loss, _ = tr.train_step(inputs.to(DEV), covers.to(DEV), alphas.to(DEV))
    assert abs(0.5 * (r0["loss"][0] + r1["loss"][0]) - loss.item()) <= 1e-6 * abs(loss.item()) + 1e-9
    # the mean of the two shards' gradients is the whole-batch gradient up to fp32 summation order
    g1 = tr.opt.flat_grad.cpu().numpy()
    np.testing.assert_array_equal(r0["grad"], r1["grad"])
    assert np.abs(r0["grad"] - g1).max() <= 2e-6 * np.abs(g1).max(), np.abs(r0["grad"] - g1).max() / np.abs(g1).max()
    nbad = ntot = 0
    for k, p in model.named_parameters():
        ref = p.detach().cpu().numpy()
        np.testing.assert_array_equal(r0["p_" + k], r1["p_" + k], err_msg=f"replicas diverged: {k}")     # bitwise identical replicas
        # the first AdamW step moves every weight by lr * g / (|g| + eps): where |g| is near eps = 1e-8 the summation-order noise of the
        # gradient changes the step, so a handful of weights may differ by a fraction of lr; everything else agrees to fp32 rounding
        # (the gradient and bitwise-replica checks above carry this test; here only: almost every weight agrees to fp32 rounding, and no
        # weight moved by more than one AdamW step, 2 * lr)
        d = np.abs(r0["p_" + k] - ref)
        assert d.max() <= 2.0e-3 * 1.05, k
        nbad += int((d > 2e-6).sum()); ntot += d.size
    assert nbad <= 1e-4 * ntot, (nbad, ntot)
    # both ranks stop after the same number of epochs, on the GLOBAL validation average (1.0, then 1.1 -> patience 1 exhausted)
    np.testing.assert_array_equal(r0["fit"], r1["fit"])
    assert r0["fit"][0] == 2 and abs(r0["fit"][1] - 1.0) < 1e-12
    np.testing.assert_allclose(r0["val"], [1.0, 1.1], atol=1e-12)
    np.testing.assert_array_equal(r0["val"], r1["val"])


def test_two_rank_step_planar_training(tmp_path):
    """The same two-rank step in the default planar training arithmetic (train_mode 'f16f8p'): replicas bitwise identical, the all-reduced
    gradient equal on both ranks and equal to the single-process whole-batch gradient up to the arithmetic (each rank scales its gradients by
    its own power of two and sums in its own order: relative L2 <= 2e-3), the epoch reduction with the range-flag slot stops both together."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    outs = [tmp_path / f"rank{r}.npz" for r in range(2)]
    procs = [subprocess.Popen([sys.executable, str(HERE / "dp_worker.py"), str(r), "2", str(port), str(outs[r]), "f16f8p"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("data-parallel ranks dead-locked (timeout)")
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    r0, r1 = np.load(outs[0]), np.load(outs[1])
    sys.path.insert(0, str(HERE))
    from dp_worker import batch
    covers, inputs, alphas = batch()
    model = gpu_model(1, "he", "f16f8p")
    assert model.train_mode == "f16f8p"
    tr = Trainer(model, loss="l1ws", lr=1e-3)
    loss, _ = tr.train_step(inputs.to(DEV), covers.to(DEV), alphas.to(DEV))
    assert abs(0.5 * (r0["loss"][0] + r1["loss"][0]) - loss.item()) <= 1e-5 * abs(loss.item())
    g1 = tr.opt.flat_grad.double().cpu().numpy()
    np.testing.assert_array_equal(r0["grad"], r1["grad"])
    rel = np.linalg.norm(r0["grad"].astype(np.float64) - g1) / np.linalg.norm(g1)
    assert rel <= 2e-3, rel
    for k, _ in model.named_parameters():
        np.testing.assert_array_equal(r0["p_" + k], r1["p_" + k], err_msg=f"replicas diverged: {k}")
    np.testing.assert_array_equal(r0["fit"], r1["fit"])
    assert r0["fit"][0] == 2


def test_two_rank_sharded_evaluate(tmp_path):
    """BASELINE.json configs[3] on the one GPU of this box: two fresh processes run the real `predict_unet_sharded` over a 5-cover /
    2-stego data set in the default mode (planar 'f16f8p') -- rows split 3 + 2 (covers) and 1 + 1 (stego), results all-gathered --
    and BOTH return the single-rank table: same rows in fabrika's lexical order (src/fabrika.py:73), same columns
    (src/unet/evaluate.py:135-139), same values (an image's prediction does not depend on its batch)."""
    import pandas as pd
    from test_gpu_evaluate import _make_dataset
    from ws_unet_amd import evaluate
    data = tmp_path / "data"
    data.mkdir()
    _make_dataset(data)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    outs = [str(tmp_path / f"rank{r}") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, str(HERE / "eval_worker.py"), str(r), "2", str(port), str(data), outs[r]],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("sharded-evaluate ranks dead-locked (timeout)")
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    model = gpu_model(2, "he", None, drop_rate=0.)
    assert model.mode == DEFAULT_MODE
    ref_c = evaluate.predict_unet_cover(data, model=model, progress_on=False)              # the reference-shaped per-image loop
    ref_s = evaluate.predict_unet_stego(data, model=model, stego_method="LSBR")
    assert ref_c["name"].tolist() == ["images/1.png", "images/10.png", "images/22.png", "images/3.png", "images/7.png"]
    for r in range(2):
        got_c, got_s = pd.read_csv(outs[r] + ".cover.csv"), pd.read_csv(outs[r] + ".stego.csv")
        assert list(got_c.columns) == ["name", "height", "width", "beta_hat", "l1"], list(got_c.columns)
        assert got_c["name"].tolist() == ref_c["name"].tolist()
        np.testing.assert_allclose(got_c["beta_hat"].to_numpy(float), ref_c["beta_hat"].to_numpy(float), atol=2e-5)
        np.testing.assert_allclose(got_c["l1"].to_numpy(float), ref_c["l1"].to_numpy(float), atol=2e-5)
        assert got_s["name"].tolist() == ref_s["name"].tolist() and set(ref_s.columns) <= set(got_s.columns)
        np.testing.assert_allclose(got_s["beta_hat"].to_numpy(float), ref_s["beta_hat"].to_numpy(float), atol=2e-5)
        np.testing.assert_allclose(got_s["l1"].to_numpy(float), ref_s["l1"].to_numpy(float), atol=2e-5)
    a, b = pd.read_csv(outs[0] + ".cover.csv"), pd.read_csv(outs[1] + ".cover.csv")
    np.testing.assert_array_equal(a["beta_hat"].to_numpy(), b["beta_hat"].to_numpy())      # every rank holds the same table


def test_two_rank_sharded_evaluate_one_rank_overflows(tmp_path):
    """only rank 0's shard trips the +-448 range flag (its model switches to 'bf16x3s' by its own first-forward look in the
    middle of the pass).  The end-of-pass decision is collective on EVERY rank whatever its local mode: nobody hangs in a mismatched
    all-reduce, both ranks finish in 'bf16x3s' and both hold the single-rank 'bf16x3s' table."""
    import pandas as pd
    from test_gpu_evaluate import _make_dataset
    from ws_unet_amd import evaluate
    data = tmp_path / "data"
    data.mkdir()
    _make_dataset(data)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    outs = [str(tmp_path / f"rank{r}") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, str(HERE / "eval_worker.py"), str(r), "2", str(port), str(data), outs[r], "default", "0"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("sharded-evaluate ranks dead-locked (mismatched collective)")
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    model = gpu_model(2, "he", "bf16x3s", drop_rate=0.)
    ref_c = evaluate.predict_unet_cover(data, model=model, progress_on=False)
    for r in range(2):
        got_c = pd.read_csv(outs[r] + ".cover.csv")
        assert got_c["name"].tolist() == ref_c["name"].tolist()
        np.testing.assert_allclose(got_c["beta_hat"].to_numpy(float), ref_c["beta_hat"].to_numpy(float), atol=2e-6)
        np.testing.assert_allclose(got_c["l1"].to_numpy(float), ref_c["l1"].to_numpy(float), atol=2e-6)
