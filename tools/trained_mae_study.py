"""How the default inference mode's accuracy margin holds on weights that went through training (VERDICT r03 missing #5; the reference ships no
UNet checkpoint: .MISSING_LARGE_BLOBS:7-12).  unet_2 from the PyTorch-default-like formula init is trained by this package's own loop in three
published-style configurations with FRESH synthetic images every step, then 4 unseen 512x512 images go through every inference mode and through the fp32
CPU oracle: MAE of the [0,1] output per mode.  One JSON line per configuration.  python tools/trained_mae_study.py [steps]"""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from ws_unet_amd import formula, ops
from ws_unet_amd.model import get_model
from ws_unet_amd.trainer import Trainer
from oracle import unet_ref

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda", 0)
u8 = formula.synthetic_images(4, 512, 512, seed=987)
x = torch.from_numpy(u8.astype(np.float32) / np.float32(255.))[:, None]
for name, loss, drop, covers_only, lr in (("LSBR-style: l1ws, cover / LSBR alpha 0.4 pairs", "l1ws", None, False, 1e-3),
                                          ("dropout-style: l1, covers only, drop_rate 0.1", "l1", 0.1, True, 1e-3),
                                          ("LSBR-style at the published learning rate 1e-4", "l1ws", None, False, 1e-4)):
    m = get_model("unet_2", in_channels=1, out_channels=1, channel=[0], drop_rate=drop, mode=None)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in formula.formula_state_dict(2, "default").items()})
    m = m.to(dev)
    tr = Trainer(m, loss=loss, lr=lr)
    t0 = time.perf_counter()
    first = last = None
    for s in range(steps):
        cov = formula.synthetic_images(8, 256, 256, seed=10_000 + s)                      # fresh images every step
        st = cov if covers_only else np.stack([formula.lsbr_embed(c, 0.4, seed=s * 8 + i) if i % 2 else c for i, c in enumerate(cov)])
        covers = ops.u8_to_unit(torch.from_numpy(cov).to(dev))[:, None].contiguous()
        inputs = ops.u8_to_unit(torch.from_numpy(st).to(dev))[:, None].contiguous()
        alphas = torch.tensor([0.0 if covers_only or i % 2 == 0 else 0.4 for i in range(8)], device=dev)
        l, _ = tr.train_step(inputs, covers, alphas)
        if s == 0:
            first = float(l.item())
    last = float(l.item())
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    del m, tr
    with torch.no_grad():
        ref = unet_ref.unet_forward(x.clone(), {k: v.float() for k, v in sd.items()}, 2)
    res = {"config": name, "steps": steps, "loss_first": first, "loss_last": last, "train_s": time.perf_counter() - t0,
           "oracle_output_std": float(ref.std()), "oracle_l1_vs_input_gray_levels": float((ref - x).abs().mean() * 255), "mae": {}}
    for md in ("f16f4p", "f16f8p", "bf16x3s", "bf16", "f32"):
        mm = get_model("unet_2", in_channels=1, out_channels=1, channel=[0], drop_rate=0. if drop else None, mode=md)
        mm.load_state_dict(sd)
        mm = mm.to(dev)
        with torch.no_grad():
            y = mm(x.to(dev)).cpu()
        res["mae"][md] = float((y - ref).abs().mean())
        res.setdefault("max", {})[md] = float((y - ref).abs().max())
        assert mm.mode == md, (md, mm.mode)
        del mm
    print(json.dumps(res), flush=True)
