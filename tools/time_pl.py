"""A/B of the two f16f8 forward conv kernels on single layers, same process, interleaved rounds (guide rule 24):
conv3x3_kernel<F16F8> on NHWC-chunk storage vs the persistent LDS-DMA kernel on planar storage (conv3x3_pl).
    python tools/time_pl.py [rounds]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from ws_unet_amd import _lib
if __name__ == "__main__" and len(sys.argv) > 2:   # A/B build variants: python tools/time_pl.py 2 libwsu_prio5.so
    _lib.LIB_PATH = Path(_lib.LIB_PATH).parent / sys.argv[2]
from ws_unet_amd import ops

M = ops.mode_id("f16f8")


def enc_nhwc(x):            # (N,H,W,C) fp32 -> 3 B/element NHWC-chunk storage
    n, h, w, c = x.shape
    xc = x.reshape(n, h, w, c // 16, 16)
    hi = xc.to(torch.float16)
    lo = ((xc - hi.float()) * 4096.0).clamp(-448, 448).to(torch.float8_e4m3fn)
    raw = torch.cat([hi.view(torch.uint8).reshape(n, h, w, c // 16, 32), lo.view(torch.uint8)], -1)
    return raw.reshape(n, h, w, c * 3).contiguous().view(torch.float32)


def enc_planar(x):          # (N,H,W,C) fp32 -> planar storage
    n, h, w, c = x.shape
    xc = x.reshape(n, h, w, c // 16, 16).permute(0, 3, 1, 2, 4).contiguous()
    hi = xc.to(torch.float16)
    lo = ((xc - hi.float()) * 4096.0).clamp(-448, 448).to(torch.float8_e4m3fn)
    hb = hi.view(torch.uint8).reshape(n, c // 16, h, w, 2, 16)
    return torch.stack([hb[..., 0, :], hb[..., 1, :], lo.view(torch.uint8)], dim=2).contiguous().view(torch.float32)


def bench(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def run(cin, cout, hw, n=32, c2=0, rounds=2, pool=False, head=False):
    g = torch.Generator(device="cuda").manual_seed(cin * 1000 + hw)
    def act(c):
        base = [torch.randn(1, hw, hw, c, device="cuda", generator=g).clamp_min(0) for _ in range(4)]
        return (torch.cat([enc_nhwc(b) for b in base] * (n // 4)), torch.cat([enc_planar(b) for b in base] * (n // 4)))
    a1, p1 = act(cin - c2)
    a2, p2 = act(c2) if c2 else (None, None)
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.zeros(cout, device="cuda")
    wp = ops.pack_conv3x3(w, M)
    hwt = torch.randn(1, 64, 1, 1, device="cuda") * 0.1 if head else None
    hb = torch.zeros(1, device="cuda") if head else None
    if head:
        old = lambda: ops.conv3x3_head(a1, a2, wp, b, hwt, hb, M)
        new = lambda: ops.conv3x3_pl(p1, p2, wp, b, cout, head_w=hwt, head_b=hb, want_y=False)
    else:
        old = lambda: ops.conv3x3(a1, a2, wp, b, cout, M, pool=pool)
        new = lambda: ops.conv3x3_pl(p1, p2, wp, b, cout, pool=pool)
    to, tn = [], []
    for _ in range(rounds):
        to.append(bench(old)); tn.append(bench(new))
    fl = 2 * 9 * cin * cout * n * hw * hw
    print(f"{Path(_lib.LIB_PATH).name}: cin={cin} cout={cout} hw={hw} concat={c2} pool={int(pool)} head={int(head)}: old {min(to):.0f} us ({fl / min(to) / 1e6:.0f} TF/s)  "
          f"planar {min(tn):.0f} us ({fl / min(tn) / 1e6:.0f} TF/s)  speed-up {min(to) / min(tn):.2f}x", flush=True)


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    run(64, 64, 512, rounds=rounds); run(64, 128, 256, rounds=rounds); run(128, 128, 256, rounds=rounds, pool=True)
    run(256, 256, 128, rounds=rounds); run(256, 128, 256, c2=128, rounds=rounds); run(128, 64, 512, c2=64, rounds=rounds)
    run(64, 64, 512, rounds=rounds, head=True); run(64, 64, 512, rounds=rounds, pool=True)
