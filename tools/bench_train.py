"""Train-step timing (BASELINE.json configs[2]): unet_2 fwd + L1WS loss + bwd + AdamW on a synthetic batch of
512x512 cover/stego pairs.  Prints one JSON line with images/s (whole job) and the per-kernel time split of rank 0.
Usage: python tools/bench_train.py [--batch 16] [--steps 5] [--size 512] [--train-mode f16f8p|bf16x3|f32]
Data parallel (BASELINE configs[4], e.g. 8 x batch 8 @ 1024): one process per GPU, `--batch` is per rank, the flat gradient
bucket is all-reduced over RCCL every step:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_train.py --batch 8 --size 1024"""
import argparse, json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from ws_unet_amd import formula, ops, parallel
from ws_unet_amd.model import get_model
from ws_unet_amd.trainer import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--train-mode", default="f16f8p")
a = ap.parse_args()
rank, world = parallel.init_from_env()
dev = torch.device("cuda", torch.cuda.current_device())
m = get_model("unet_2", in_channels=1, out_channels=1, channel=[0], drop_rate=None, mode="f32" if a.train_mode == "f32" else "f16f8p")
m.load_state_dict({k: torch.from_numpy(v) for k, v in formula.formula_state_dict(2, "default").items()})
m = m.to(dev); m.train_mode = a.train_mode
cov = formula.synthetic_images(a.batch, a.size, a.size, seed=5 + rank)      # every rank its own shard of the global batch
st = np.stack([formula.lsbr_embed(c, 0.4, seed=i) if i % 2 else c for i, c in enumerate(cov)])
covers = ops.u8_to_unit(torch.from_numpy(cov).to(dev))[:, None].contiguous()
inputs = ops.u8_to_unit(torch.from_numpy(st).to(dev))[:, None].contiguous()
alphas = torch.tensor([0.4 if i % 2 else 0.0 for i in range(a.batch)], device=dev)
tr = Trainer(m, loss="l1ws", lr=1e-4)
for _ in range(2):
    tr.train_step(inputs, covers, alphas)
torch.cuda.synchronize()
if world > 1:
    torch.distributed.barrier()
timer = ops.KernelTimer(); ops.set_timer(timer)
t0 = time.perf_counter()
for _ in range(a.steps):
    loss, _ = tr.train_step(inputs, covers, alphas)
torch.cuda.synchronize()
if world > 1:
    torch.distributed.barrier()
dt = time.perf_counter() - t0
ops.set_timer(None)
if world > 1:
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = t.item()
ks = {k: round(v["total_ms"] / a.steps, 3) for k, v in timer.summary().items()}
if rank == 0:
    print(json.dumps({"metric": "train images/s (unet_2 fwd+L1WS+bwd+AdamW)", "value": world * a.batch * a.steps / dt, "ms_per_step": dt / a.steps * 1e3,
                      "n_gpus": world, "batch_per_gpu": a.batch, "size": a.size, "train_mode": a.train_mode, "loss": loss.item(),
                      "tflops_algorithmic": world * a.batch * a.steps * 606e9 * (a.size / 512) ** 2 / dt / 1e12,
                      "kernel_ms_per_step": ks, "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30}))
if world > 1:
    torch.distributed.destroy_process_group()
