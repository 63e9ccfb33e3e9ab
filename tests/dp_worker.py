"""One data-parallel rank of tests/test_gpu_dp.py (a fresh process per rank: `python dp_worker.py RANK WORLD PORT OUT.npz [MODE]`).
Two of these share the box's single GPU and talk over the gloo backend on CUDA tensors, which exercises the same
Trainer / parallel code path that RCCL serves on a multi-GPU node."""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))


def batch(n=4, size=64, seed=71):
    from ws_unet_amd import formula
    cov_u8 = formula.synthetic_images(n, size, size, seed=seed)
    st_u8 = np.stack([formula.lsbr_embed(c, 0.4, seed=seed + i) if i % 2 else c for i, c in enumerate(cov_u8)])
    covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None]
    inputs = torch.from_numpy(st_u8.astype(np.float32) / np.float32(255.))[:, None]
    alphas = torch.tensor([0.4 if i % 2 else 0.0 for i in range(n)])
    return covers, inputs, alphas


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    mode = sys.argv[5] if len(sys.argv) > 5 else "f32"
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "RANK": str(rank), "WORLD_SIZE": str(world)})
    from ws_unet_amd import parallel
    from ws_unet_amd.trainer import Trainer
    from gpu_util import gpu_model, DEV
    parallel.init_from_env("gloo")
    model = gpu_model(1, "he", mode)
    if rank != 0:                                             # replicas start different: Trainer's ONE flat broadcast must repair it
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.01 * rank)
    tr = Trainer(model, loss="l1ws", lr=1e-3, patience=1)
    res = {}
    covers, inputs, alphas = batch()
    per = inputs.shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)
    loss, _ = tr.train_step(inputs[sl].to(DEV), covers[sl].to(DEV), alphas[sl].to(DEV))
    res["loss"] = np.array([loss.item()])
    res["grad"] = (tr.opt.flat_grad * (1.0 / world)).cpu().numpy()       # the all-reduced bucket with the 1/world the optimiser folds in
    for k, p in model.named_parameters():
        res["p_" + k] = p.detach().cpu().numpy()

    # fit(): per-rank validation losses that would make the ranks disagree about early stopping (rank 0 keeps improving, rank 1
    # degrades faster); the epoch sums are all-reduced, so every rank must see the global average 1.0, 1.1, 1.2 ... and stop together
    sched = {"i": 0}

    def fake_eval(inputs_, covers_, alphas_):
        e = sched["i"]
        sched["i"] += 1
        val = 1.0 - 0.1 * e if rank == 0 else 1.0 + 0.3 * e
        tr._last_l1 = None
        return torch.tensor(val, device=DEV), torch.zeros_like(inputs_)

    tr.eval_step = fake_eval
    loader = [(inputs[sl], (covers[sl], alphas[sl]))]
    best = tr.fit(loader, loader, num_epochs=6)
    res["fit"] = np.array([tr.epochs_run, best, tr.patience])
    res["val"] = np.array([v for _, t, v in tr.scalars if t == "val/loss"])
    np.savez(out, **res)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
