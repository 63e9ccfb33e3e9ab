"""Train step of the planar path with the backward matrix kernels in both `products` (include/wsu.h WSU_PRODUCTS_*), interleaved on one box:
python tools/ab_products.py [batch size rounds].  Prints ms per step (HIP events over `steps` steps) and the per-kernel totals of one step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ws_unet_amd import losses, ops
from ws_unet_amd.model import get_model

if __name__ == "__main__":
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = torch.device("cuda", 0)
    m = get_model("unet_2", in_channels=1, out_channels=1, channel=[0], drop_rate=None, mode="f16f8p").to(dev)
    gen = torch.Generator(device="cpu").manual_seed(0)
    x = torch.rand((batch, 1, size, size), generator=gen).to(dev)
    cov = torch.rand((batch, 1, size, size), generator=gen).to(dev)
    alphas = torch.full((batch,), 0.4, device=dev)
    crit = losses.L1WSLoss()

    def step():
        m.zero_grad(set_to_none=True)
        crit(m(x), (cov, alphas), x).backward()

    def timed(steps=4):
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    for r in range(rounds):
        for pr in ("f16f8", "f16"):
            m.train_products = pr
            print(f"round {r} products={pr:6s} fwd+bwd {timed():7.2f} ms", flush=True)
    for pr in ("f16f8", "f16"):
        m.train_products = pr
        step(); torch.cuda.synchronize()
        t = ops.KernelTimer(); ops.set_timer(t)
        try:
            step(); torch.cuda.synchronize()
        finally:
            ops.set_timer(None)
        print(pr, {k: round(v["total_ms"], 3) for k, v in t.summary().items()}, flush=True)
