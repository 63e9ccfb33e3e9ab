"""Error of mode 'f16f8' against the exact fp32 mode / the CPU oracle, next to bf16x3 (python tools/diag_f16f8.py)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import torch  # noqa: E402
from gpu_util import DEV, gpu_model, images01, oracle_forward  # noqa: E402

for ns, size, init in [(2, 64, "he"), (2, 64, "default"), (0, 64, "he"), (3, 64, "he"), (2, 512, "he")]:
    _, x = images01(2, size, size, seed=11)
    with torch.no_grad():
        ref = gpu_model(ns, init, "f32")(x.to(DEV)).cpu()
        out = {md: gpu_model(ns, init, md)(x.to(DEV)).cpu() for md in ("bf16x3s", "f16f8", "bf16")}
    line = f"unet_{ns} {size}x{size} init={init}:"
    for md, y in out.items():
        line += f"  {md}: mean {((y - ref).abs().mean().item()):.3g} max {((y - ref).abs().max().item()):.3g}"
    if size <= 64:
        line += f"  | f32 vs oracle max {(ref - oracle_forward(x, ns, init)).abs().max().item():.3g}"
    print(line, flush=True)
