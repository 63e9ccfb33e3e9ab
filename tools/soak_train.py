"""Stability soak of the planar training path: N AdamW steps on a fixed synthetic batch (the loss must fall, no step may be skipped by the finite
guard, the range flag must stay clear).  python tools/soak_train.py [steps] [batch] [size] [train_mode] [train_products]
With WSU_SOAK_JSON=path the loss of every step is written there (tools/soak_compare.py puts trajectories of several arithmetics side by side)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ws_unet_amd import formula, ops
from ws_unet_amd.model import get_model
from ws_unet_amd.trainer import Trainer

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    tm = sys.argv[4] if len(sys.argv) > 4 else None
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)                                       # the same initial weights in every run
    m = get_model("unet_2", in_channels=1, out_channels=1, channel=[0], drop_rate=None, mode="f16f8p")
    if tm:
        m.train_mode = tm
    if len(sys.argv) > 5:
        m.train_products = sys.argv[5]
    m = m.to(dev)                                              # PyTorch default init, as a training run starts
    cov = formula.synthetic_images(batch, size, size, seed=5)
    st = np.stack([formula.lsbr_embed(c, 0.4, seed=i) if i % 2 else c for i, c in enumerate(cov)])
    covers = ops.u8_to_unit(torch.from_numpy(cov).to(dev))[:, None].contiguous()
    inputs = ops.u8_to_unit(torch.from_numpy(st).to(dev))[:, None].contiguous()
    alphas = torch.tensor([0.4 if i % 2 else 0.0 for i in range(batch)], device=dev)
    tr = Trainer(m, loss="l1ws", lr=1e-3)
    losses, every = [], []
    for i in range(steps):
        loss, _ = tr.train_step(inputs, covers, alphas)
        every.append(loss)
        if i % max(1, steps // 10) == 0 or i == steps - 1:
            losses.append((i, float(loss.item())))
    if os.environ.get("WSU_SOAK_JSON"):
        import json
        with open(os.environ["WSU_SOAK_JSON"], "w") as fh:
            json.dump({"train_mode": m.train_mode, "train_products": getattr(m, "train_products", None), "batch": batch, "size": size,
                       "loss": [float(v.item()) for v in every]}, fh)
    print("train_mode", m.train_mode, "products", getattr(m, "train_products", None), "losses", [(i, round(v, 5)) for i, v in losses], "skipped", tr.skipped_steps(), "range_exceeded", m.range_exceeded(), flush=True)
