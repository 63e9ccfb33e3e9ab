"""Diagnostic: per-phase timeline of conv3x3 v1 workgroups from in-kernel s_memtime stamps (WSU_CONV_ABLATE=512).
python tools/stamp_conv.py mode cin cout hw batch"""
import sys, os, ctypes
os.environ["WSU_CONV_ABLATE"] = str(512 | int(os.environ.get("EXTRA_ABLATE", "0")))
os.environ["WSU_CONV_IMPL"] = "v1"
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from ws_unet_amd import ops, _lib
mode, cin, cout, hw, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
m = ops.mode_id(mode); dt = ops.act_dtype(m)
x = torch.rand(n, hw, hw, ops.store_channels(cin, m), device="cuda").to(dt)
w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
wp = ops.pack_conv3x3(w, m); b = torch.zeros(cout, device="cuda")
for _ in range(3):
    y = ops.conv3x3(x, None, wp, b, cout, m)
torch.cuda.synchronize()
lib = _lib.load()
NB, NS = 2048, 32
buf = (ctypes.c_ulonglong * (NB * NS))()
lib.wsu_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.wsu_debug_read_stamps(buf, NB) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(NB, NS).astype(np.int64)
nch = cin // (32 if mode == "bf16" else 16)
t0 = st[:, 0].min()
print(f"{mode} cin={cin} cout={cout} hw={hw} n={n}: s_memtime ticks (100 MHz? -> see ratio); first-wave start spread {st[:512,0].max()-t0}")
rel = st - st[:, :1]
def med(a): return float(np.median(a))
print("load issue          :", med(rel[:, 1]))
for c in range(min(nch, 6)):
    A, B, C, D = 2 + 4 * c, 3 + 4 * c, 4 + 4 * c, 5 + 4 * c
    prev = rel[:, 1] if c == 0 else rel[:, 5 + 4 * (c - 1)]
    print(f"chunk {c}: wait+barrierA {med(rel[:,A]-prev):8.0f}  commit+barrierB {med(rel[:,B]-rel[:,A]):8.0f}  prefetch issue {med(rel[:,C]-rel[:,B]):8.0f}  MFMA {med(rel[:,D]-rel[:,C]):8.0f}")
last = 5 + 4 * (min(nch, 6) - 1)
print(f"epilogue: acc->LDS+barrier {med(rel[:,26]-rel[:,last]):8.0f}  stores {med(rel[:,27]-rel[:,26]):8.0f}   total/block {med(rel[:,27]):8.0f}")
# concurrency: blocks sorted by start; how many blocks start within the first block's lifetime
print("block lifetimes: median", med(rel[:, 27]), " p10", float(np.percentile(rel[:, 27], 10)), " p90", float(np.percentile(rel[:, 27], 90)))

real = (st[:, 31] - st[:, 30]).astype(np.float64)          # 100 MHz ticks
cyc = (st[:, 27] - st[:, 0]).astype(np.float64)
ok = real > 0
print(f"in-kernel clock (s_memtime / s_memrealtime x 100 MHz): median {np.median(cyc[ok] / real[ok]) * 0.1:.3f} GHz")
