"""Race screen for the fused decoder entry (ops.conv3x3_up_q, csrc/conv3x3_qu.hip: its LDS ring, run-time vmcnt waits and barriers): many launches of
several shapes -- few / many tiles per workgroup, ragged tiles, several output blocks -- every result compared bitwise with the first one, other work
queued in between.  python tools/stress_qu.py [iterations]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import torch
from ws_unet_amd import ops
from gpu_util import planar_q_encode
_num = [a for a in sys.argv[1:] if a.isdigit()]
iters = int(_num[0]) if _num else 100
g = torch.Generator(device="cuda").manual_seed(7)
bad_total = 0
for (n, hl, wl, cl, c2, cout) in [(8, 32, 48, 128, 64, 64), (4, 64, 64, 256, 128, 128), (16, 24, 40, 64, 32, 64), (2, 128, 128, 128, 64, 64), (32, 16, 16, 512, 256, 256),
                                  (1, 5, 7, 32, 16, 64), (32, 128, 128, 256, 128, 128)]:
    cup = cl // 2
    xl = planar_q_encode(torch.randn(n, cl, hl, wl, device="cuda", generator=g).clamp_min(0))
    xs = planar_q_encode(torch.randn(n, c2, 2 * hl, 2 * wl, device="cuda", generator=g).clamp_min(0))
    w3 = torch.randn(cout, cup + c2, 3, 3, device="cuda", generator=g) * (2.0 / (9 * (cup + c2))) ** 0.5
    wt = torch.randn(cl, cup, 2, 2, device="cuda", generator=g) * (1.0 / cl) ** 0.5
    ws, wl_, bias = ops.pack_conv3x3_up(w3, wt, torch.randn(cup, device="cuda", generator=g) * 0.1, torch.randn(cout, device="cuda", generator=g) * 0.1)
    hw = 4 * hl * wl
    ref = ops.conv3x3_up_q(xl, xs, ws, wl_, bias, cout).data[:, :, :48 * hw].clone()
    torch.cuda.synchronize()
    bad = 0
    for it in range(iters):
        out = ops.conv3x3_up_q(xl, xs, ws, wl_, bias, cout)
        if it % 3 == 0:
            torch.randn(1 << 20, device="cuda").sum()
        bad += 0 if torch.equal(out.data[:, :, :48 * hw], ref) else 1
    bad_total += bad
    print(f"shape n={n} low {hl}x{wl} cl={cl} c2={c2} cout={cout}: {bad} of {iters} launches differ from the first", flush=True)
print("TOTAL mismatching launches:", bad_total)

# the optional fused first layer of the default mode (conv3x3_q kernel variant F1: loader waves computing e11 into the LDS input slots)
if "--f1" in sys.argv:
    bad_total = 0
    for (n, h, w, cout) in [(8, 128, 160, 64), (2, 512, 512, 64), (16, 64, 64, 128), (32, 256, 256, 64)]:
        x = torch.rand(n, 1, h, w, device="cuda", generator=g)
        w1, b1 = torch.randn(64, 1, 3, 3, device="cuda", generator=g) * 0.5, torch.randn(64, device="cuda", generator=g) * 0.1
        w2 = torch.randn(cout, 64, 3, 3, device="cuda", generator=g) * (2.0 / (9 * 64)) ** 0.5
        b2 = torch.randn(cout, device="cuda", generator=g) * 0.1
        wp = ops.pack_conv3x3_f4(w2)
        y0, p0 = ops.conv3x3_q_fused_first(x, w1, b1, wp, b2, cout)
        r0, q0 = y0.data[:, :, :48 * h * w].clone(), p0.data[:, :, :12 * h * w].clone()
        bad = 0
        for it in range(iters):
            y, p = ops.conv3x3_q_fused_first(x, w1, b1, wp, b2, cout)
            if it % 3 == 0:
                torch.randn(1 << 20, device="cuda").sum()
            bad += 0 if (torch.equal(y.data[:, :, :48 * h * w], r0) and torch.equal(p.data[:, :, :12 * h * w], q0)) else 1
        bad_total += bad
        print(f"fused first layer n={n} {h}x{w} cout={cout}: {bad} of {iters} launches differ from the first", flush=True)
    print("TOTAL mismatching launches (fused first layer):", bad_total)
