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

from gpu_util import DEV, gpu_model
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
        d = np.abs(r0["p_" + k] - ref)
        assert d.max() <= 2.1e-3, k
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
