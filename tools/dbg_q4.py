import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, torch.nn.functional as F
from ws_unet_amd import ops
from gpu_util import DEV, planar_encode, planar_decode
from test_gpu_planar import _q4_blocks, _fp4
n, h, w, cin, cout = 1, 16, 32, 64, 64
g = torch.Generator().manual_seed(7)
x = planar_decode(planar_encode(torch.relu(torch.randn((n, cin, h, w), generator=g))))
wgt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
b = torch.zeros(cout)
xp = F.pad(x, (1, 1, 1, 1), mode="reflect")
xh = xp.half().float(); xr = ((xp - xh) * 4096).to(torch.float8_e4m3fn).float() / 4096
wh = wgt.half().float()
xc4, xr4 = _q4_blocks(xh, xr * 2048.0, 1)
wc4, wr4 = _q4_blocks(wh, (wgt - wh) * 2048.0, 1)
main = F.conv2d(xh.double(), wh.double())
t1 = F.conv2d(xc4.double(), wr4.double()); t2 = F.conv2d(xr4.double(), wc4.double())
wp = ops.pack_conv3x3_f4(wgt.to(DEV))
y = planar_decode(ops.conv3x3_pl(planar_encode(x).to(DEV), None, wp, b.to(DEV), cout, relu=False, x_residual=2)).double()
d = y - main
def corr(a, b): return float((a * b).sum() / (a.norm() * b.norm()))
print("|main|", float(main.norm()), "|d_gpu|", float(d.norm()), "|t1|", float(t1.norm()), "|t2|", float(t2.norm()), "|t1+t2|", float((t1 + t2).norm()))
print("corr(d, t1+t2)", corr(d, t1 + t2), "corr(d,t1)", corr(d, t1), "corr(d,t2)", corr(d, t2))
for name, cand in (("t1+t2", t1 + t2), ("t1", t1), ("t2", t2), ("2(t1+t2)", 2 * (t1 + t2)), ("(t1+t2)/2", (t1 + t2) / 2)):
    print(name, "resid", float((d - cand).norm()))
# per-tap analysis: which taps contribute
for tap in range(9):
    wm = torch.zeros_like(wr4); wm[:, :, tap // 3, tap % 3] = 1
    tt = F.conv2d(xc4.double(), (wr4 * wm).double()) + F.conv2d(xr4.double(), (wc4 * wm).double())
    print("tap", tap, "corr with d", round(corr(d, tt), 3), "proj", round(float((d * tt).sum() / (tt * tt).sum()), 3))
print("---- centre tap only")
wg2 = torch.zeros_like(wgt); wg2[:, :, 1, 1] = wgt[:, :, 1, 1]
wh = wg2.half().float()
wc4, wr4 = _q4_blocks(wh, (wg2 - wh) * 2048.0, 1)
main = F.conv2d(xh.double(), wh.double()); t12 = F.conv2d(xc4.double(), wr4.double()) + F.conv2d(xr4.double(), wc4.double())
y = planar_decode(ops.conv3x3_pl(planar_encode(x).to(DEV), None, ops.pack_conv3x3_f4(wg2.to(DEV)), b.to(DEV), cout, relu=False, x_residual=2)).double()
d = y - main
err = (d - t12)[0].norm(dim=0) / t12[0].norm(dim=0).clamp_min(1e-12)      # per pixel
torch.set_printoptions(linewidth=250, precision=2, sci_mode=False)
print((err > 0.05).int())
errc = (d - t12)[0].norm(dim=(1, 2)) / t12[0].norm(dim=(1, 2))
print("per output channel rel err", errc)
print("---- variants (centre tap only)")
def run(xv, wv, tag):
    xp = F.pad(xv, (1, 1, 1, 1), mode="reflect")
    xh = xp.half().float(); xr = ((xp - xh) * 4096).to(torch.float8_e4m3fn).float() / 4096
    wh = wv.half().float()
    xc4, xr4 = _q4_blocks(xh, xr * 2048.0, 1); wc4, wr4 = _q4_blocks(wh, (wv - wh) * 2048.0, 1)
    main = F.conv2d(xh.double(), wh.double()); t1 = F.conv2d(xc4.double(), wr4.double()); t2 = F.conv2d(xr4.double(), wc4.double())
    y = planar_decode(ops.conv3x3_pl(planar_encode(xv).to(DEV), None, ops.pack_conv3x3_f4(wv.to(DEV)), b.to(DEV), cout, relu=False, x_residual=2)).double()
    d = y - main
    print(tag, "|d|", float(d.norm()), "|t1|", float(t1.norm()), "|t2|", float(t2.norm()), "corr(d,t1)", round(corr(d, t1), 3) if t1.norm() > 0 else None,
          "corr(d,t2)", round(corr(d, t2), 3) if t2.norm() > 0 else None, "|d-(t1+t2)|", float((d - t1 - t2).norm()))
    return d, t1, t2
run(x.half().float(), wg2, "x exact f16 (t2 = 0)")
run(x, wg2.half().float(), "w exact f16 (t1 = 0)")
xs = torch.full_like(x, 1.0) + 0.0004 * torch.rand(x.shape, generator=g)
xs = planar_decode(planar_encode(xs))
run(xs, wg2, "x ~ 1 (all block scales equal)")
