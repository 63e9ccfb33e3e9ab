"""Transposed-conv layers of unet_2 at batch 32 in mode f16f8 (WSU_CONVT_TILE=2 selects the 2x32 tile): python tools/time_convt.py"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from ws_unet_amd import ops

m = ops.mode_id(sys.argv[1] if len(sys.argv) > 1 else "f16f8")
for cin, cout, hw in ((256, 128, 128), (128, 64, 256)):
    x = torch.rand(32, hw, hw, ops.store_channels(cin, m), device="cuda")
    w = torch.randn(cin, cout, 2, 2, device="cuda") * 0.05
    b = torch.zeros(cout, device="cuda")
    wp = ops.pack_convt2x2(w, m)
    for _ in range(3):
        ops.convt2x2(x, wp, b, cout, m)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.convt2x2(x, wp, b, cout, m)
    e.record(); torch.cuda.synchronize()
    print(f"convT {cin}->{cout} @{hw} tile={os.environ.get('WSU_CONVT_TILE', '4')}: {s.elapsed_time(e) * 100:.0f} us", flush=True)
