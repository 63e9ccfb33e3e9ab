"""Timing of single conv3x3 layers through the PLANAR persistent kernel (conv3x3_pl) with an alternative build of libwsu -- the probes of
profiles/r02/conv3x3_units_probe.md repeated on the kernel that ships (VERDICT r02 next #3c: they had been timed on the NHWC kernel only).
`make -C ws_unet_amd/csrc probes` -> libwsu_plprobeN.so (WSU_PROBE = 2: the fp8 instructions read their registers as fp4, 3: no cross terms
= plain f16, 4: fp6); results are WRONG for N != 0.  The 15-unit variant (one cross term) is the product library's `x_residual=False`.

    python tools/probe_units_pl.py [libwsu_plprobe2.so] [--xres0]        one process per library (a process loads one libwsu)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from ws_unet_amd import _lib
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
if libs:
    _lib.LIB_PATH = Path(_lib.LIB_PATH).parent / libs[0]
XRES = "--xres0" not in sys.argv
Q4 = False                                # (the fp4 variant moved to csrc/conv3x3_q.hip in round 4; its probe build: make qprobe16)
ZEROS = "--zeros" in sys.argv                # all-zero activations and weights: the same instruction stream with (almost) no switching in the data paths --
                                             # what the kernel would do if it were not power-managed (profiles/r03/conv3x3_pl_last_step.md)
from ws_unet_amd import ops
from time_pl import enc_planar

M = ops.mode_id("f16f8")


def run(cin, cout, hw, n=32, c2=0, pool=False):
    g = torch.Generator(device="cuda").manual_seed(cin * 1000 + hw)
    def act(c):
        parts = [enc_planar(torch.randn(1, hw, hw, c, device="cuda", generator=g).clamp_min(0)) for _ in range(4)]
        return torch.cat(parts * (n // 4))
    x1 = act(cin - c2)
    x2 = act(c2) if c2 else None
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5
    if ZEROS:
        x1.zero_(); w.zero_()
        if x2 is not None: x2.zero_()
    b = torch.zeros(cout, device="cuda")
    wp = ops.pack_conv3x3_f4(w) if Q4 else ops.pack_conv3x3(w, M)
    fn = lambda: ops.conv3x3_pl(x1, x2, wp, b, cout, pool=pool, x_residual=XRES)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10)
    fl = 2 * 9 * cin * cout * n * hw * hw
    print(f"{Path(_lib.LIB_PATH).name}{' q4' if Q4 else '' if XRES else ' x_residual=0'}{' zeros' if ZEROS else ''}: cin={cin} cout={cout} hw={hw} concat={c2} pool={int(pool)}: {best * 1e3:.0f} us  {fl / best / 1e9:.0f} TFLOP/s algorithmic", flush=True)


run(64, 64, 512, pool=True); run(64, 128, 256); run(128, 128, 256, pool=True); run(256, 256, 128); run(256, 128, 256, c2=128); run(128, 64, 512, c2=64)
