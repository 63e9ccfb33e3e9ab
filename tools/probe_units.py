"""Timing of single conv3x3 layers in mode f16f8 with an alternative build of libwsu (timing-only probes of the matrix section:
`make -C ws_unet_amd/csrc probes` -> libwsu_probeN.so, wsu_device.h WSU_PROBE; results are WRONG for N != 0).
Inputs are properly encoded f16f8 activations of post-ReLU-like data (random bytes would decode to Inf/NaN f16 values).

    python tools/probe_units.py [libwsu_probe1.so]        one process per library (a process loads one libwsu)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from ws_unet_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = Path(_lib.LIB_PATH).parent / sys.argv[1]
from ws_unet_amd import ops

M = ops.mode_id("f16f8")


def encode_f16f8(x):
    """(N,H,W,C) fp32 on the device -> the 3-byte-per-element storage of include/wsu.h as a float32-typed buffer."""
    n, h, w, c = x.shape
    xc = x.reshape(n, h, w, c // 16, 16)
    hi = xc.to(torch.float16)
    lo = ((xc - hi.float()) * 4096.0).clamp(-448, 448).to(torch.float8_e4m3fn)
    raw = torch.cat([hi.view(torch.uint8).reshape(n, h, w, c // 16, 32), lo.view(torch.uint8)], -1)
    return raw.reshape(n, h, w, c * 3).contiguous().view(torch.float32)


def run(cin, cout, hw, n=32, c2=0):
    g = torch.Generator(device="cuda").manual_seed(cin * 1000 + hw)
    def act(c):
        parts = [encode_f16f8(torch.randn(1, hw, hw, c, device="cuda", generator=g).clamp_min(0)) for _ in range(4)]
        return torch.cat(parts * (n // 4))
    x1 = act(cin - c2)
    x2 = act(c2) if c2 else None
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.zeros(cout, device="cuda")
    wp = ops.pack_conv3x3(w, M)
    for _ in range(3):
        ops.conv3x3(x1, x2, wp, b, cout, M)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.conv3x3(x1, x2, wp, b, cout, M)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    fl = 2 * 9 * cin * cout * n * hw * hw
    print(f"{Path(_lib.LIB_PATH).name}: cin={cin} cout={cout} hw={hw} concat={c2}: {ms * 1e3:.0f} us  {fl / ms / 1e9:.0f} TFLOP/s algorithmic", flush=True)


for rep in range(2):
    run(64, 64, 512); run(64, 128, 256); run(128, 128, 256); run(256, 256, 128); run(256, 128, 256, c2=128); run(128, 64, 512, c2=64)
