"""CPU tests of the multi-rank logic (world_size 2, gloo): batch-sharded evaluate, result gathering in
fabrika order, flat-bucket gradient all-reduce, parameter broadcast."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ws_unet_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world)})
    r, w = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world) == parallel.world_info()
    res = {}
    # 1. sharded evaluate: 7 rows over 2 ranks, fake predictor = deterministic function of the row
    rows = [f"images/{k}.png" for k in (1, 10, 11, 2, 20, 3, 7)]
    seen = []

    def predict(chunk):
        seen.extend(chunk)
        return torch.tensor([[len(c) + 0.5, float(c.split("/")[1].split(".")[0])] for c in chunk])

    table = parallel.evaluate_sharded(rows, predict, batch_size=2)
    res["table"] = table.numpy()
    res["seen"] = seen
    # 2. flat gradient bucket: sum all-reduce, scale = 1/world
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    scale = parallel.allreduce_flat_(flat)
    res["flat"], res["scale"] = flat.numpy(), scale
    # 3. parameter broadcast from rank 0
    torch.manual_seed(100 + rank)
    lin = torch.nn.Linear(4, 3)
    parallel.broadcast_parameters(lin)
    res["w"] = lin.weight.detach().numpy().copy()
    # 4. more ranks than rows: empty shard must not dead-lock
    t2 = parallel.evaluate_sharded(["only"], lambda c: torch.tensor([[1.0, 2.0]] * len(c)))
    res["t2"] = t2.numpy()
    # 4b. ... also when the predictor returns k != 2 columns (the empty rank learns k from the others) and for indices beyond 2^24
    t3 = parallel.evaluate_sharded(["only"], lambda c: torch.tensor([[1.0, 2.0, 3.0, 4.0, 5.0]] * len(c)))
    res["t3"] = t3.numpy()
    big = (1 << 24) + 1
    mine = [big] if rank == 0 else [big + 2]
    t4 = parallel.gather_rows(mine, torch.tensor([[float(rank + 1)]]), big + 3)
    res["t4"] = np.array([t4[big, 0].item(), t4[big + 2, 0].item(), float(torch.isnan(t4[big + 1, 0]))])
    # 5. epoch sums: per-rank validation results differ, every rank must derive the SAME global averages and the same stop decision
    best, patience, stops, vals = float("inf"), 2, [], []
    for epoch in range(6):
        per_rank_loss = (1.0 - 0.1 * epoch) if rank == 0 else (1.0 + 0.3 * epoch)      # rank 0 improves, rank 1 degrades faster
        nimg = 3 if rank == 0 else 5                                                    # unequal shards: averages weigh by image count
        sums = torch.tensor([per_rank_loss * nimg, 0.0, 0.0, float(nimg), 1.0], dtype=torch.float64)
        tot = parallel.reduce_epoch_sums_(sums)
        val = (tot[0] / tot[3]).item()
        vals.append(val)
        best, patience, stop = parallel.early_stop_update(val, best, patience, 2)
        stops.append(stop)
        if stop:
            break
    res["vals"], res["stop_epoch"], res["best"] = np.array(vals), len(vals), best
    # 6. one flat broadcast
    flat = torch.full((7,), float(rank + 5))
    parallel.broadcast_flat_(flat)
    res["flatb"] = flat.numpy()
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_indices_cover_and_balance():
    for n in (0, 1, 5, 8, 13):
        for world in (1, 2, 3, 8):
            parts = [parallel.shard_indices(n, r, world) for r in range(world)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    # single process: evaluate_sharded degenerates to a plain ordered loop
    t = parallel.evaluate_sharded([3, 1, 2], lambda c: torch.tensor([[float(v), 0.0] for v in c]), batch_size=2)
    assert t[:, 0].tolist() == [3.0, 1.0, 2.0]


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows = [f"images/{k}.png" for k in (1, 10, 11, 2, 20, 3, 7)]
    expect = np.array([[len(c) + 0.5, float(c.split("/")[1].split(".")[0])] for c in rows], dtype=np.float32)
    for r in (0, 1):
        np.testing.assert_array_equal(out[r]["table"], expect)            # every rank holds the full table, in row order
        np.testing.assert_array_equal(out[r]["flat"], np.arange(10, dtype=np.float32) * 3)
        assert out[r]["scale"] == 0.5
        np.testing.assert_array_equal(out[r]["t2"], np.array([[1.0, 2.0]], dtype=np.float32))
        np.testing.assert_array_equal(out[r]["t3"], np.array([[1.0, 2.0, 3.0, 4.0, 5.0]], dtype=np.float32))
        np.testing.assert_array_equal(out[r]["t4"], [1.0, 2.0, 1.0])
        np.testing.assert_array_equal(out[r]["flatb"], np.full(7, 5.0, dtype=np.float32))
    # global validation average = (3 * (1 - 0.1 e) + 5 * (1 + 0.3 e)) / 8 = 1 + 0.15 e: never improves after epoch 0 -> stop after 3 epochs
    np.testing.assert_allclose(out[0]["vals"], [1.0, 1.15, 1.3], rtol=1e-12)
    np.testing.assert_array_equal(out[0]["vals"], out[1]["vals"])
    assert out[0]["stop_epoch"] == out[1]["stop_epoch"] == 3 and out[0]["best"] == out[1]["best"] == 1.0
    assert out[0]["seen"] == rows[:4] and out[1]["seen"] == rows[4:]       # contiguous shards, no overlap
    np.testing.assert_array_equal(out[0]["w"], out[1]["w"])
