"""WS payload estimate throughput on resident uint8 planes (src/ws/estimate.py `attack`, batched on the device):
UNet predictor pass(es) + wsu_ws_attack; also the statistic kernel alone and with an in-kernel linear predictor.
python tools/bench_ws_attack.py [--batch 32] [--steps 10] [--correct-bias]"""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from ws_unet_amd import filters, formula, ops  # noqa: E402
from ws_unet_amd.model import get_model  # noqa: E402
from ws_unet_amd.ws import estimate  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--mode", default=None, help="precision mode of the predictor (default: the package default, f16f4p)")
ap.add_argument("--correct-bias", action="store_true")
a = ap.parse_args()
dev = "cuda"
m = get_model("unet_2", in_channels=1, out_channels=1, channel=[0], drop_rate=0., mode=a.mode)
m.load_state_dict({k: torch.from_numpy(v) for k, v in formula.formula_state_dict(2, "he").items()})
m = m.to(dev)
est = estimate.UNetEstimator(m)
x = torch.from_numpy(formula.synthetic_images(a.batch, 512, 512, seed=5)).to(dev)
AVG = filters.NAMED_FILTERS_2D["AVG"]


def timed(fn, steps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


t_unet = timed(lambda: estimate._stat(x, est, AVG, 1, a.correct_bias), a.steps)
kb = filters.get_filter_estimator(filter_name="KB", flatten=False)
t_kb = timed(lambda: estimate._stat(x, kb, AVG, 1, a.correct_bias), 50)
y = torch.rand(a.batch, 512, 512, device=dev)
t_stat = timed(lambda: ops.ws_attack(x, y, mean_filter=AVG, weighted=1), 50)
npx = a.batch * 512 * 512
print(json.dumps({"metric": "WS attack images/s (UNet predictor + weighted statistic)", "value": a.batch / t_unet,
                  "ms_per_batch": t_unet * 1e3, "batch": a.batch, "mode": m.mode, "correct_bias": a.correct_bias,
                  "filter_KB_images_per_s": a.batch / t_kb, "stat_call_us": t_stat * 1e6,
                  "stat_GBps_algorithmic": npx * 5 / t_stat / 1e9}))
