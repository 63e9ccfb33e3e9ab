"""Fused first layer (e11 + e12 [+pool]) per storage format: python tools/time_first_layer.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from ws_unet_amd import ops

x = torch.rand(32, 1, 512, 512, device="cuda")
w1 = torch.randn(64, 1, 3, 3, device="cuda") * 0.5
b1 = torch.zeros(64, device="cuda")
w2 = torch.randn(64, 64, 3, 3, device="cuda") * 0.06
b2 = torch.zeros(64, device="cuda")
for rep in range(2):
    for mode in ("bf16x3", "bf16x3s", "f16f8"):
        m = ops.mode_id(mode)
        wp = ops.pack_conv3x3(w2, ops.first_layer_weight_mode(m))
        for pool in (True, False):
            for _ in range(3):
                ops.conv3x3_fused_first(x, w1, b1, wp, b2, 64, m, pool=pool)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ops.conv3x3_fused_first(x, w1, b1, wp, b2, 64, m, pool=pool)
            e.record(); torch.cuda.synchronize()
            print(f"{mode:8s} pool={pool}: {s.elapsed_time(e) * 100:.0f} us", flush=True)
