"""Where does the input gradient differ from the CPU oracle?  (interior / border ring / corners, exact-fp32 model)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np, torch
from gpu_util import gpu_model, DEV
from ws_unet_amd import formula, losses, ops
from oracle import unet_ref, losses_ref
for size in (64, 128):
    u8 = formula.synthetic_images(2, size, size, seed=11)
    x0 = torch.from_numpy(u8.astype(np.float32) / np.float32(255.))[:, None]
    cov = x0.clone(); al = torch.tensor([0.4, 0.0])
    ref = unet_ref.build_ref(2, formula.formula_state_dict(2, "he"))
    xr = x0.clone().requires_grad_(True)
    losses_ref.l2_loss(ref(xr), (cov, al)).backward()
    m = gpu_model(2, "he", "f32")
    xd = x0.to(DEV).requires_grad_(True)
    losses.L2Loss()(m(xd), (cov.to(DEV), al.to(DEV)), xd).backward()
    a, b = xd.grad.cpu().double(), xr.grad.double()
    def rel(sl): return float((a[sl] - b[sl]).norm() / b[sl].norm())
    I = (slice(None), slice(None), slice(2, -2), slice(2, -2))
    print(size, "all", rel(tuple([slice(None)] * 4)), "interior", rel(I))
    for r in (0, 1, 2, size - 3, size - 2, size - 1):
        print("  row", r, rel((slice(None), slice(None), slice(r, r + 1), slice(2, -2))), " col", r, rel((slice(None), slice(None), slice(2, -2), slice(r, r + 1))))
    print("  corner 2x2", rel((slice(None), slice(None), slice(0, 2), slice(0, 2))))
    # the kernel alone against autograd of the first layer
    g = torch.from_numpy(formula.formula_tensor("dg", (2, 64, size, size), 1.0))
    w = torch.from_numpy(formula.formula_tensor("dw", (64, 1, 3, 3), 0.5))
    xi = x0.clone().requires_grad_(True)
    unet_ref.conv3x3_reflect(xi, w, None).backward(g)
    got = ops.conv3x3_first_bwd_data(g.permute(0, 2, 3, 1).contiguous().to(DEV), w.to(DEV)).cpu().double()
    print("  kernel alone: rel L2", float((got - xi.grad.double()).norm() / xi.grad.double().norm()))
