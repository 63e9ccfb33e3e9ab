import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gpu_util import DEV, gpu_model
def rel_l2(a,b): return float((a-b).norm()/b.norm().clamp_min(1e-30))
for ns in (3,4):
    model = gpu_model(ns, "he", "f16f8p")
    x = torch.rand((2, 1, 64, 64), generator=torch.Generator().manual_seed(5)).to(DEV)
    tgt = torch.rand((2, 1, 64, 64), generator=torch.Generator().manual_seed(6)).to(DEV)
    res = {}
    for tm in ("f32", "f16f8p", "bf16x3"):
        model.train_mode = tm
        model.zero_grad()
        ((model(x) - tgt) ** 2).mean().backward()
        res[tm] = {k: p.grad.detach().double().cpu() for k, p in model.named_parameters()}
    for k in res["f32"]:
        print(ns, k, "f16f8p %.2e" % rel_l2(res["f16f8p"][k], res["f32"][k]), "bf16x3 %.2e" % rel_l2(res["bf16x3"][k], res["f32"][k]), flush=True)
