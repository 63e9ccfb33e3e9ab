"""Train-step timing of one training arithmetic: python tools/time_train.py [train_mode] [batch] [size] (bench.py's train_step leg)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

if __name__ == "__main__":
    tm = sys.argv[1] if len(sys.argv) > 1 else None
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    res = bench.train_step_leg(torch.device("cuda", 0), batch, size, train_mode=tm)
    print(json.dumps(res))
    if os.environ.get("WSU_TIME_TRAIN_LAUNCHES"):
        # one more step with every launch listed (kernel, ms, algorithmic TFLOP/s, GB/s from the op's own flop / byte counts)
        import numpy as np
        from ws_unet_amd import formula, ops
        from ws_unet_amd.model import get_model
        from ws_unet_amd.trainer import Trainer
        dev = torch.device("cuda", 0)
        m = get_model("unet_2", in_channels=1, out_channels=1, channel=[0], drop_rate=None, mode="f16f8p")
        if tm:
            m.train_mode = tm
        m.load_state_dict({k: torch.from_numpy(v) for k, v in formula.formula_state_dict(2, "default").items()})
        m = m.to(dev)
        cov = formula.synthetic_images(batch, size, size, seed=5)
        covers = ops.u8_to_unit(torch.from_numpy(cov).to(dev))[:, None].contiguous()
        alphas = torch.zeros(batch, device=dev)
        tr = Trainer(m, loss="l1ws", lr=1e-4)
        for _ in range(2):
            tr.train_step(covers, covers, alphas)
        timer = ops.KernelTimer()
        ops.set_timer(timer)
        tr.train_step(covers, covers, alphas)
        torch.cuda.synchronize()
        ops.set_timer(None)
        for k, meta, s, e in timer.records:
            ms = s.elapsed_time(e)
            print(f"{k:32s} {ms:8.3f} ms  {meta.get('flops', 0) / ms / 1e9:8.1f} TFLOP/s  {meta.get('bytes', 0) / ms / 1e6:8.1f} GB/s", file=sys.stderr)
