import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from ws_unet_amd import ops, _lib
from gpu_util import DEV, planar_encode, planar_decode
from test_gpu_planar import _q4_blocks, _fp4
lib = _lib.load()
fn = lib.wsu_debug_q4_encode
fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]; fn.restype = ctypes.c_int
TAB = torch.tensor([0, .5, 1, 1.5, 2, 3, 4, 6, -0., -.5, -1, -1.5, -2, -3, -4, -6])
g = torch.Generator().manual_seed(3)
h, w = 8, 16
x = planar_decode(planar_encode(torch.randn((1, 16, h, w), generator=g) * torch.exp2(torch.randint(-4, 5, (1, 1, h, w), generator=g).float())))
xe = planar_encode(x).to(DEV)                       # (1, 1, 3, h, w, 4) float32-typed
q = torch.zeros(h * w * 16, dtype=torch.uint8, device=DEV); sb = torch.zeros(h * w, dtype=torch.uint8, device=DEV)
assert fn(xe.data_ptr(), q.data_ptr(), sb.data_ptr(), h * w, None) == 0
torch.cuda.synchronize()
qb = q.cpu().view(h * w, 16).long(); sbc = sb.cpu().long()
nib = torch.stack([qb & 15, qb >> 4], dim=-1).reshape(h * w, 32)      # nibble i of the granule
vals = TAB[nib] * torch.exp2(sbc.float() - 127)[:, None]
xh = x.half().float(); xr = ((x - xh) * 4096).to(torch.float8_e4m3fn).float() / 4096
c4, r4 = _q4_blocks(xh, xr * 2048.0, 1)
c4 = c4[0].permute(1, 2, 0).reshape(h * w, 16); r4 = r4[0].permute(1, 2, 0).reshape(h * w, 16) * 2048.0
print("scale bytes gpu", sbc[:8].tolist(), "expected exps", (torch.floor(torch.log2(xh[0].abs().amax(0).reshape(-1).clamp_min(2.0**-14))) - 1 + 127)[:8].tolist())
print("copy part max diff", float((vals[:, :16] - c4).abs().max()), "residual part max diff", float((vals[:, 16:] - r4).abs().max()))
print("pixel 0 gpu copy", vals[0, :16].tolist()); print("pixel 0 emu copy", c4[0].tolist())
print("pixel 0 gpu res ", vals[0, 16:].tolist()); print("pixel 0 emu res ", r4[0].tolist())
# packed weights
wgt = torch.randn((64, 16, 3, 3), generator=g) * 0.1
wp = ops.pack_conv3x3_f4(wgt.to(DEV)).cpu()
sl = wp[:28672]
tap, co = 4, 5
gran = sl[(tap * 3 + 2) * 1024 + co * 16:(tap * 3 + 2) * 1024 + co * 16 + 16].long()
nibw = torch.stack([gran & 15, gran >> 4], dim=-1).reshape(32)
sw = int(sl[27648 + tap * 64 + co])
wv = wgt[co, :, tap // 3, tap % 3]; wh = wv.half().float()
E = int(torch.floor(torch.log2(wh.abs().max())).item()) - 1
print("w scale byte", sw, "expected", E + 127 - 11)
print("w res gpu", (TAB[nibw[:16]] * 2.0 ** E / 2048).tolist()); print("w res true", (wv - wh).tolist())
print("w copy gpu", (TAB[nibw[16:]] * 2.0 ** E).tolist()); print("w copy true", wh.tolist())
