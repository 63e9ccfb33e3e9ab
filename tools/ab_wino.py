"""Direct vs Winograd F(2,3) conv3x3 timing, mode bf16x3: python tools/ab_wino.py cin cout hw batch"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from ws_unet_amd import ops
cin, cout, hw, n = (int(v) for v in sys.argv[1:5])
m = ops.mode_id("bf16x3")
x = torch.rand(n, hw, hw, cin, device="cuda")
w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
b = torch.zeros(cout, device="cuda")
wp, wpw = ops.pack_conv3x3(w, m), ops.pack_conv3x3_wino(w)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 10

td = timed(lambda: ops.conv3x3(x, None, wp, b, cout, m))
tw = timed(lambda: ops.conv3x3_wino(x, None, wpw, b, cout))
fl = 2 * 9 * cin * cout * n * hw * hw
print(f"cin={cin} cout={cout} hw={hw} n={n}: direct {td*1e3:.0f} us ({fl/td/1e9:.0f} TF/s)  wino {tw*1e3:.0f} us ({fl/tw/1e9:.0f} TF/s)  x{td/tw:.3f}")
