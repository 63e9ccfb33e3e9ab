"""Timing-only ablation of conv3x3_kernel (results are wrong when WSU_CONV_ABLATE != 0).
Times one layer shape per process: python tools/ablate_conv.py mode cin cout hw batch"""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from ws_unet_amd import ops
mode, cin, cout, hw, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
m = ops.mode_id(mode)
dt = ops.act_dtype(m)
x = torch.rand(n, hw, hw, ops.store_channels(cin, m), device="cuda").to(dt)
w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
b = torch.zeros(cout, device="cuda")
wp = ops.pack_conv3x3(w, m)
for _ in range(3):
    y = ops.conv3x3(x, None, wp, b, cout, m)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    y = ops.conv3x3(x, None, wp, b, cout, m)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
fl = 2 * 9 * cin * cout * n * hw * hw
print(f"{mode} cin={cin} cout={cout} hw={hw} n={n} ablate={os.environ.get('WSU_CONV_ABLATE','0')} waves={os.environ.get('WSU_CONV_WAVES','-')}: {ms*1e3:.0f} us  {fl/ms/1e9:.0f} TFLOP/s")
