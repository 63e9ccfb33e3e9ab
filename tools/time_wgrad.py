"""Per-layer timing of the planar weight-gradient kernel at the production shapes: python tools/time_wgrad.py  (WSU_WGRAD_ABLATE=1|2 for the
timing-only variants: no matrix section / no staging after the first tile)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ws_unet_amd import ops

SHAPES = [(64, 512, 64, 0, 64), (64, 256, 64, 0, 128), (64, 256, 128, 0, 128), (64, 128, 128, 0, 256), (64, 128, 256, 0, 256),
          (64, 256, 128, 128, 128), (64, 512, 64, 64, 64)]

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    for n, s, c1, c2, cout in SHAPES:
        g = torch.randint(0, 2 ** 31 - 1, ops.planar_shape(n, cout, s, s), dtype=torch.int32, device=dev).view(torch.float32)
        g.view(torch.int32).bitwise_and_(0x3BFF3BFF)                # finite f16 halves, small e4m3 bytes
        x1 = g[:, : c1 // 16].contiguous() if c1 == cout else torch.zeros(ops.planar_shape(n, c1, s, s), device=dev)
        x2 = torch.zeros(ops.planar_shape(n, c2, s, s), device=dev) if c2 else None
        for _ in range(2):
            ops.conv3x3_pl_bwd_weight(g, x1, x2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv3x3_pl_bwd_weight(g, x1, x2)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        fl = 2.0 * 9 * (c1 + c2) * cout * n * s * s
        print(f"wgrad {c1}+{c2}->{cout} @{s}: {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
