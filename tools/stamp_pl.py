"""Phase stamps of conv3x3_pl (WSU_PL_STAMP=1): shader cycles per chunk step spent waiting for the DMA, in the barrier, issuing the
next DMA, in the matrix section, and per tile in the epilogue; in-kernel clock = s_memtime / s_memrealtime x 100 MHz.
    python tools/stamp_pl.py cin cout hw [c2] [--q4] [--pool]        (--q4: the fp4 variant; its stamps build takes -DWSU_PL_EPO=0 -- the stamps sit in the plain step loop)"""
import ctypes, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from ws_unet_amd import _lib
_lib.LIB_PATH = Path(_lib.LIB_PATH).parent / "libwsu_plstamp.so"          # `make -C ws_unet_amd/csrc probes`
from ws_unet_amd import ops
sys.path.insert(0, str(Path(__file__).resolve().parent))
from time_pl import enc_planar          # noqa: E402  (runs nothing: time_pl guards its main)
Q4 = False; POOL = "--pool" in sys.argv          # (the fp4 variant moved to csrc/conv3x3_q.hip in round 4: it carries no stamps)
pos = [a for a in sys.argv[1:] if not a.startswith("--")]
cin, cout, hw = int(pos[0]), int(pos[1]), int(pos[2])
c2 = int(pos[3]) if len(pos) > 3 else 0
n = 32
M = ops.mode_id("f16f8")
g = torch.Generator(device="cuda").manual_seed(1)
def act(c):
    base = [enc_planar(torch.randn(1, hw, hw, c, device="cuda", generator=g).clamp_min(0)) for _ in range(4)]
    return torch.cat(base * (n // 4))
p1 = act(cin - c2); p2 = act(c2) if c2 else None
w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5
wp = ops.pack_conv3x3_f4(w) if Q4 else ops.pack_conv3x3(w, M); b = torch.zeros(cout, device="cuda")
for _ in range(20):
    ops.conv3x3_pl(p1, p2, wp, b, cout, pool=POOL, x_residual=1)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (256 * 8))()
lib.wsu_debug_read_pl_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.wsu_debug_read_pl_stamps(buf, 256) == 0
allrows = np.array(buf[:], dtype=np.float64).reshape(128, 2, 8)
per_wave_bar = allrows[64:128]           # blocks 64-127: [block][0][wave 0-7] barrier wait of the matrix waves, [block][1][0-3] of the fp4 loaders (accumulated cycles)
per_wave_mma = allrows[32:64, 0]         # blocks 32-63: matrix section time per matrix wave
allrows = allrows[:32]
ld = allrows[:, 1][allrows[:, 1, 7] > 0]
s = allrows[:, 0][allrows[:, 0, 7] > 0]
J = s[:, 7]
print(f"cin={cin} cout={cout} hw={hw} c2={c2}: workgroups {len(s)}, steps/WG {np.median(J):.0f}")
if len(ld) and Q4:
    print("  loader wave 0 (cycles/step): " + "  ".join(f"{nm} {np.median(ld[:, k] / ld[:, 7]):.0f}" for k, nm in ((3, "barrier"), (4, "issue W(j+1) + IN(j+2)"), (2, "wait IN(j+1)"), (5, "derive"), (6, "wait W(j+1)"))))
elif len(ld):
    print(f"  loader wave: wait vmcnt {np.median(ld[:, 2] / ld[:, 7]):.0f}  barrier {np.median(ld[:, 3] / ld[:, 7]):.0f}  DMA issue {np.median(ld[:, 4] / ld[:, 7]):.0f} cycles/step")
print(f"  in-kernel clock {np.median(s[:, 0] / s[:, 1]) * 0.1:.3f} GHz; kernel {np.median(s[:, 1]) / 100:.0f} us; cycles/step {np.median(s[:, 0] / J):.0f}")
for k, name in ((2, "wait vmcnt"), (3, "barrier"), (4, "DMA issue (+tile plan)"), (5, "matrix section")):
    print(f"  {name:24s} {np.median(s[:, k] / J):8.0f} cycles/step  ({100 * np.median(s[:, k] / s[:, 0]):.1f} %)")
ntile = J / ((cin) // 16)
print(f"  {'epilogue':24s} {np.median(s[:, 6] / ntile):8.0f} cycles/tile  ({100 * np.median(s[:, 6] / s[:, 0]):.1f} %)")
Jn = float(np.median(J))
print("  barrier wait per wave (cycles/step): matrix " + " ".join(f"{np.median(per_wave_bar[:, 0, w]) / Jn:.0f}" for w in range(8))
      + (" | loaders " + " ".join(f"{np.median(per_wave_bar[:, 1, w]) / Jn:.0f}" for w in range(4)) if Q4 else ""))
print("  matrix section per wave (cycles/step): " + " ".join(f"{np.median(per_wave_mma[:, w]) / Jn:.0f}" for w in range(8)))
