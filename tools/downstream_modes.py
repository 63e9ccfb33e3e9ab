"""What the inference mode's rounding does to the WS statistics (beta_hat, l1 of evaluate.py:125-132) -- python tools/downstream_modes.py [images]
For `images` synthetic 512x512 covers and their LSBr (alpha 0.4) stegos: beta_hat / l1 in modes f32 (exact), f16f8p, f16f4p (default) on the gate's weights."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ws_unet_amd import evaluate, formula
from ws_unet_amd.model import get_model

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda", 0)
    sd = {k: torch.from_numpy(v) for k, v in formula.formula_state_dict(2, "he").items()}
    cov = formula.synthetic_images(n, 512, 512, seed=2024)
    st = np.stack([formula.lsbr_embed(c, 0.4, seed=i) for i, c in enumerate(cov)])
    res = {}
    for mode in ("f32", "f16f8p", "f16f4p"):
        m = get_model("unet_2", in_channels=1, out_channels=1, channel=[0], drop_rate=0., mode=mode)
        m.load_state_dict(sd)
        m = m.to(dev)
        rows = []
        for arr in (cov, st):
            for i in range(0, n, 32):
                b, l = evaluate.predict_u8_batch(torch.from_numpy(arr[i:i + 32]).to(dev), m)
                rows.append(torch.stack([b, l], 1).cpu())
        res[mode] = torch.cat(rows).double().numpy()
    ref = res["f32"]
    print(f"{2 * n} images (covers + LSBr 0.4 stegos), unet_2 on the gate's weights; beta_hat spread of the exact mode: covers {ref[:n, 0].std():.4f}, stegos {ref[n:, 0].std():.4f}; "
          f"mean stego - cover beta_hat {ref[n:, 0].mean() - ref[:n, 0].mean():.4f}")
    for mode in ("f16f8p", "f16f4p"):
        d = np.abs(res[mode] - ref)
        print(f"  {mode}: |beta_hat - exact| max {d[:, 0].max():.2e} mean {d[:, 0].mean():.2e};  |l1 - exact| max {d[:, 1].max():.2e} mean {d[:, 1].mean():.2e} (l1 in 0..255 units, values ~{ref[:, 1].mean():.1f})")
