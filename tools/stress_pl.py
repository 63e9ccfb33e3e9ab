"""Race screen for conv3x3_pl (e4m3 cross terms, planar e4m3-residual tensors) or, with --q4, conv3x3_q (fp4 cross terms, planar Q tensors: the
default inference mode): many launches of several shapes, every result compared bitwise with the first one.  Prints the number of
mismatching launches per shape.  python tools/stress_pl.py [iterations] [--q4]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from ws_unet_amd import ops
from time_pl import enc_planar, enc_nhwc
M = ops.mode_id("f16f8")
_num = [a for a in sys.argv[1:] if a.isdigit()]
iters = int(_num[0]) if _num else 100
Q4 = "--q4" in sys.argv                      # csrc/conv3x3_q.hip
g = torch.Generator(device="cuda").manual_seed(7)
bad_total = 0
for (n, h, w, c1, c2, cout, pool, head) in [(8, 64, 96, 64, 0, 64, True, False), (4, 128, 128, 128, 128, 128, False, False), (16, 48, 80, 64, 0, 64, False, True),
                                            (2, 256, 256, 64, 0, 128, True, False), (32, 32, 32, 256, 0, 256, False, False)]:
    x1 = torch.randn(n, h, w, c1, device="cuda", generator=g).clamp_min(0)
    x2 = torch.randn(n, h, w, c2, device="cuda", generator=g).clamp_min(0) if c2 else None
    wt = torch.randn(cout, c1 + c2, 3, 3, device="cuda", generator=g) * (2.0 / (9 * (c1 + c2))) ** 0.5
    b = torch.randn(cout, device="cuda", generator=g) * 0.1
    wp = ops.pack_conv3x3_f4(wt) if Q4 else ops.pack_conv3x3(wt, M)
    if Q4:                                   # a planar Q input: written by the first-layer-style producer path of the library itself
        sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
        from gpu_util import planar_q_encode
        p1, p2 = planar_q_encode(x1.permute(0, 3, 1, 2)), (planar_q_encode(x2.permute(0, 3, 1, 2)) if c2 else None)
    else:
        p1, p2 = enc_planar(x1), (enc_planar(x2) if c2 else None)
    hw_ = torch.randn(1, 64, 1, 1, device="cuda", generator=g) * 0.2 if head else None
    hb = torch.zeros(1, device="cuda") if head else None
    def run():
        conv = ops.conv3x3_q if Q4 else ops.conv3x3_pl
        if head:
            return conv(p1, p2, wp, b, cout, head_w=hw_, head_b=hb, want_y=True)
        r = conv(p1, p2, wp, b, cout, pool=pool)
        return r if pool else (r,)
    raw = lambda t: t.data[:, :, :48 * t.h * t.w] if isinstance(t, ops.PlanarQ) else t          # (a Q tensor's scale-plane padding is never written)
    ref = [raw(t).clone() for t in run()]
    torch.cuda.synchronize()
    bad = 0
    for it in range(iters):
        out = run()
        if it % 3 == 0:                          # other work in between: different timing of the next launch
            torch.randn(1 << 20, device="cuda").sum()
        ok = all(torch.equal(raw(a_), b_) for a_, b_ in zip(out, ref))
        bad += 0 if ok else 1
    bad_total += bad
    print(f"shape n={n} {h}x{w} c1={c1} c2={c2} cout={cout} pool={pool} head={head}: {bad} of {iters} launches differ from the first", flush=True)
print("TOTAL mismatching launches:", bad_total)
