import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from gpu_util import DEV, gpu_model, images01
for n, size in ((2, 1024), (1, 2048), (3, 516), (64, 512)):
    _, x = images01(min(n, 4), size, size, seed=3)
    if n > 4: x = x.repeat(n // 4, 1, 1, 1)
    with torch.no_grad():
        y = gpu_model(2, "he", "f16f8")(x.to(DEV))
        r = gpu_model(2, "he", "bf16x3")(x.to(DEV))
    d = (y - r).abs()
    print(n, size, "f16f8 vs bf16x3: mean %.3g max %.3g" % (d.mean().item(), d.max().item()), "peak mem GB %.1f" % (torch.cuda.max_memory_allocated() / 2**30), flush=True)
