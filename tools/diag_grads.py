"""Diagnostic (not a test): gradient error of the GPU backward vs an fp64 CPU oracle, next to the error of the
fp32 CPU oracle vs the same fp64 truth.  Usage: python tools/diag_grads.py [nsteps] [size] [f32|bf16x3]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np
import torch
from gpu_util import gpu_model, DEV
from ws_unet_amd import formula, losses
from oracle import unet_ref, losses_ref

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
mode = sys.argv[3] if len(sys.argv) > 3 else "f32"          # 'f32' (exact) or 'bf16x3' (split-bf16 train arithmetic)
cov_u8 = formula.synthetic_images(2, size, size, seed=11)
st_u8 = cov_u8.copy(); st_u8[0] = formula.lsbr_embed(cov_u8[0], 0.4, seed=5)
covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None]
inputs = torch.from_numpy(st_u8.astype(np.float32) / np.float32(255.))[:, None]
alphas = torch.tensor([0.4, 0.0])

def oracle(dtype):
    m = unet_ref.build_ref(ns, formula.formula_state_dict(ns, "he")).to(dtype)
    out = m(inputs.to(dtype))
    loss = losses_ref.l1ws_loss(out, (covers.to(dtype), alphas.to(dtype)), inputs.to(dtype))
    loss.backward()
    return {k: p.grad.double() for k, p in m.named_parameters()}, loss.item()

g64, l64 = oracle(torch.float64)
g32, l32 = oracle(torch.float32)
model = gpu_model(ns, "he", mode)
out = model(inputs.to(DEV))
loss = losses.L1WSLoss()(out, (covers.to(DEV), alphas.to(DEV)), inputs.to(DEV))
loss.backward()
print(f"loss fp64 {l64:.9f} fp32-cpu {l32:.9f} gpu {loss.item():.9f}")
print(f"{'param':18s} {'|g|max':>10s} {'cpu32 relL2':>12s} {'gpu relL2':>12s} {'cpu32 max/scale':>16s} {'gpu max/scale':>14s}")
for k, p in model.named_parameters():
    t = g64[k]; a = g32[k]; b = p.grad.double().cpu()
    n = t.norm().item(); s = t.abs().max().item()
    print(f"{k:18s} {s:10.3e} {(a - t).norm().item() / n:12.3e} {(b - t).norm().item() / n:12.3e} "
          f"{(a - t).abs().max().item() / s:16.3e} {(b - t).abs().max().item() / s:14.3e}")
