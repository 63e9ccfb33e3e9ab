"""Timing-only ablations of conv3x3_pl (WSU_PL_ABLATE bits, csrc/conv3x3_pl.hip PlArgs.ablate; results are wrong when != 0): where a chunk
step's time goes on the product kernel.  One process per setting (the variable is read once): python tools/ablate_pl.py  (runs them all)."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
    import torch
    from ws_unet_amd import ops
    from time_pl import enc_planar
    M = ops.mode_id("f16f8")
    for (cin, cout, hw, c2, pool) in [(64, 64, 512, 0, True), (128, 128, 256, 0, False), (256, 256, 128, 0, False), (128, 64, 512, 64, False)]:
        g = torch.Generator(device="cuda").manual_seed(cin * 1000 + hw)
        act = lambda c: torch.cat([enc_planar(torch.randn(1, hw, hw, c, device="cuda", generator=g).clamp_min(0)) for _ in range(4)] * 8)
        x1 = act(cin - c2); x2 = act(c2) if c2 else None
        w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5
        wp = ops.pack_conv3x3(w, M); b = torch.zeros(cout, device="cuda")
        fn = lambda: ops.conv3x3_pl(x1, x2, wp, b, cout, pool=pool)
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
        steps = 32 * ((hw + 15) // 16) * ((hw + 31) // 32) * (cout // 64) / 256 * (cin // 16)
        print(f"ablate={os.environ.get('WSU_PL_ABLATE', '0'):>2}  {cin}->{cout}@{hw} cat={c2} pool={int(pool)}: {best * 1e3:7.0f} us  {best * 1e3 / steps:6.3f} us/step", flush=True)
else:
    for ab in ("0", "2", "8", "1", "10", "11"):
        env = dict(os.environ, WSU_PL_ABLATE=ab)
        subprocess.run([sys.executable, __file__, "--one"], env=env, check=True)
