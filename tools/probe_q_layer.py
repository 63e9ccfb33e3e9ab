"""Timing of single conv3x3_q layers (planar Q tensors, the default inference mode's 3x3 conv) on random and on all-zero operands, optionally with
an alternative build of libwsu (WSU_LIB=...): the power-management measurement of profiles/r03 (tools/probe_units_pl.py) for the round-4 kernel.
python tools/probe_q_layer.py [--zeros]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import torch
from ws_unet_amd import ops, _lib
from gpu_util import planar_q_encode
ZEROS = "--zeros" in sys.argv
g = torch.Generator(device="cuda").manual_seed(1)
for (n, s, c1, c2, cout, pool) in [(32, 512, 64, 0, 64, True), (32, 256, 128, 0, 128, False), (32, 128, 256, 0, 256, False), (32, 256, 128, 128, 128, False), (32, 512, 64, 64, 64, False)]:
    mk = (lambda c: torch.zeros((n, c, s, s), device="cuda")) if ZEROS else (lambda c: torch.randn((n, c, s, s), device="cuda", generator=g).clamp_min(0))
    x1 = planar_q_encode(mk(c1)); x2 = planar_q_encode(mk(c2)) if c2 else None
    w = torch.zeros((cout, c1 + c2, 3, 3), device="cuda") if ZEROS else torch.randn((cout, c1 + c2, 3, 3), device="cuda", generator=g) * (2.0 / (9 * (c1 + c2))) ** 0.5
    wp, b = ops.pack_conv3x3_f4(w), torch.zeros(cout, device="cuda")
    fn = lambda: ops.conv3x3_q(x1, x2, wp, b, cout, pool=pool)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * 9 * (c1 + c2) * cout * n * s * s
    print(f"{Path(_lib.LIB_PATH if not __import__('os').environ.get('WSU_LIB') else __import__('os').environ['WSU_LIB']).name}{' zeros' if ZEROS else ''}: {c1}+{c2}->{cout} @{s} pool={pool}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s = {fl / us / 1e6 / 2500:.3f} of the f16 peak", flush=True)
