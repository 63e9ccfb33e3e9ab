"""Multi-GPU layer (new capability: the reference is single-process).

One process per GPU, `torch.distributed` (backend 'nccl' = RCCL over xGMI on the MI355X node; 'gloo' in the
CPU tests).  Images are independent, so:

* evaluate / predict shards the sorted row list across ranks (no data-path collective); weights are broadcast
  once (7.45 MB) and the per-image results (index, beta_hat, l1) are gathered to every rank and re-assembled
  in the original fabrika order;
* data-parallel training replicates weights + AdamW state, splits the global batch by rank and sums ONE flat
  fp32 gradient bucket (1 861 697 elements for unet_2) with a single all-reduce per step; the 1/world scale
  is folded into the AdamW kernel.  With 7 direct xGMI links per GPU the bucket is latency-, not bandwidth-
  bound (< 0.2 ms against >= 15 ms of compute), so a single flat all-reduce is used and not overlapped.

Everything here works on tensors of any device so the N > 1 logic is covered by world_size-2 gloo tests.
"""
from __future__ import annotations

import os
from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend: str = None) -> tuple:
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_indices(n: int, rank: int, world: int) -> List[int]:
    """Contiguous, balanced shard of range(n): the first n % world ranks get one extra row."""
    q, r = divmod(n, world)
    start = rank * q + min(rank, r)
    return list(range(start, start + q + (1 if rank < r else 0)))


def broadcast_parameters(model: torch.nn.Module, src: int = 0) -> None:
    rank, world = world_info()
    if world == 1:
        return
    for p in model.parameters():
        dist.broadcast(p.data, src=src)
    if hasattr(model, "invalidate_packed"):
        model.invalidate_packed()


def allreduce_flat_(flat_grad: torch.Tensor) -> float:
    """Sum-all-reduce the flat gradient bucket in place; returns the scale (1/world) the optimiser must apply."""
    rank, world = world_info()
    if world > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return 1.0 / world


def gather_rows(local_indices: Sequence[int], local_values: torch.Tensor, total: int) -> torch.Tensor:
    """All-gather per-row result vectors.  local_values: (len(local_indices), k) float32 on the compute device.
    Returns (total, k) on every rank, rows placed at their global index (NaN where no rank reported)."""
    rank, world = world_info()
    k = local_values.shape[1] if local_values.dim() == 2 else 1
    local_values = local_values.reshape(-1, k).float()
    out = torch.full((total, k), float("nan"), dtype=torch.float32, device=local_values.device)
    if world == 1:
        out[torch.as_tensor(list(local_indices), dtype=torch.long, device=out.device)] = local_values
        return out
    # equal-sized payload per rank: [count, idx..., values...] padded to the largest shard
    cap = (total + world - 1) // world
    payload = torch.zeros(1 + cap * (1 + k), dtype=torch.float32, device=local_values.device)
    cnt = len(local_indices)
    payload[0] = cnt
    payload[1:1 + cnt] = torch.as_tensor(list(local_indices), dtype=torch.float32, device=payload.device)
    payload[1 + cap:1 + cap + cnt * k] = local_values.reshape(-1)
    gathered = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload)
    for g in gathered:
        c = int(g[0].item())
        idx = g[1:1 + c].long()
        out[idx] = g[1 + cap:1 + cap + c * k].reshape(c, k)
    return out


def evaluate_sharded(rows: Sequence, predict_batch: Callable[[Sequence], torch.Tensor], batch_size: int = 32) -> torch.Tensor:
    """Batch-sharded evaluate: every rank runs ``predict_batch(rows[i:j]) -> (j-i, k)`` on its shard of ``rows``
    (kept in the caller's order) and all ranks receive the full (len(rows), k) result table.  ``batch_size=None`` hands the rank's
    whole shard to ``predict_batch`` in one call."""
    rank, world = world_info()
    mine = shard_indices(len(rows), rank, world)
    outs = []
    if batch_size is None:                              # the callable takes the rank's whole shard (it pipelines its own chunks)
        if len(mine):
            outs.append(predict_batch([rows[j] for j in mine]))
    else:
        for i in range(0, len(mine), batch_size):
            chunk = mine[i:i + batch_size]
            outs.append(predict_batch([rows[j] for j in chunk]))
    if outs:
        local = torch.cat([o.reshape(len(o), -1).float() for o in outs])
    else:
        local = torch.zeros((0, 2), dtype=torch.float32)
    k = local.shape[1]
    if world > 1:                                       # agree on k and on a device even for empty shards
        dev = local.device if len(mine) else (torch.device("cuda", torch.cuda.current_device())
                                              if dist.get_backend() == "nccl" else torch.device("cpu"))
        local = local.to(dev)
    return gather_rows(mine, local, len(rows))
