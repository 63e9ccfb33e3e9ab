"""Timing of the fused decoder-block entry (ops.conv3x3_up_q, csrc/conv3x3_qu.hip) at unet_2's two shapes (batch 32 @ 512x512), beside the two-kernel
path it replaces (convt2x2_pl -> conv3x3_q).  Environment: WSU_QU_ABLATE (timing-only ablations, results wrong), WSU_LIB.
python tools/probe_qu_layer.py [--zeros] [--no-two]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import os
import torch
from ws_unet_amd import ops
from gpu_util import planar_q_encode, planar_encode
ZEROS = "--zeros" in sys.argv
g = torch.Generator(device="cuda").manual_seed(1)


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (name, n, sl, cl, c2, cout) in [("upconv3+d31", 32, 128, 256, 128, 128), ("upconv4+d41", 32, 256, 128, 64, 64)]:
    cup = cl // 2
    mk = (lambda c, s: torch.zeros((n, c, s, s), device="cuda")) if ZEROS else (lambda c, s: torch.randn((n, c, s, s), device="cuda", generator=g).clamp_min(0))
    xl, xs = mk(cl, sl), mk(c2, 2 * sl)
    rnd = lambda *sh: torch.zeros(sh, device="cuda") if ZEROS else torch.randn(sh, device="cuda", generator=g)
    w3 = rnd(cout, cup + c2, 3, 3) * (2.0 / (9 * (cup + c2))) ** 0.5
    wt = rnd(cl, cup, 2, 2) * (1.0 / cl) ** 0.5
    b3, bt = torch.zeros(cout, device="cuda"), torch.zeros(cup, device="cuda")
    ql, qs = planar_q_encode(xl), planar_q_encode(xs)
    w_skip, w_low, bias = ops.pack_conv3x3_up(w3, wt, bt, b3)
    us = timed(lambda: ops.conv3x3_up_q(ql, qs, w_skip, w_low, bias, cout))
    fl = 2.0 * 9 * (cup + c2) * cout * n * (2 * sl) ** 2 + 2.0 * 4 * cl * cup * n * sl * sl
    tag = f"ablate={os.environ.get('WSU_QU_ABLATE', '0')}{' zeros' if ZEROS else ''}"
    line = f"{name} [{tag}]: fused {us:8.1f} us = {fl / us / 1e6:6.1f} TFLOP/s of the two reference ops"
    if "--no-two" not in sys.argv:
        al = planar_encode(xl)
        wpt, wp3 = ops.pack_convt2x2(wt, ops.mode_id("f16f8")), ops.pack_conv3x3_f4(w3)
        t_up = timed(lambda: ops.convt2x2_pl(al, wpt, bt, cup, y_format=ops.PLANAR_Q))
        xu = ops.convt2x2_pl(al, wpt, bt, cup, y_format=ops.PLANAR_Q)
        t_cv = timed(lambda: ops.conv3x3_q(xu, qs, wp3, b3, cout))
        line += f" | two kernels {t_up:7.1f} + {t_cv:7.1f} = {t_up + t_cv:8.1f} us ({(t_up + t_cv) / us:.3f}x)"
    print(line, flush=True)
