"""Round 4: the default inference mode's 3x3 conv on planar Q tensors (csrc/conv3x3_q.hip, include/wsu.h K1q) and the producers that write
that format (conv3x3_q itself, convt2x2_pl, conv3x3_first_pl with y_format = PLANAR_Q).  The checker is a CPU restatement of the arithmetic
and of the storage format (tests/gpu_util.py: planar_q_*); every call goes through the C ABI.  Replaces unet.py:141-189 like conv3x3_pl."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import (DEV, fp4_values, planar_decode, planar_encode, planar_q_decode, planar_q_encode, planar_q_parts)

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent


def _blocks_to_nchw(t):                                    # (n, chunk, h, w, 16) -> (n, c, h, w)
    n, nch, h, w, _ = t.shape
    return t.permute(0, 1, 4, 2, 3).reshape(n, nch * 16, h, w)


def _conv3x3_q_ref(x, w, b):
    """The arithmetic of conv3x3_q on the CPU, fp64 accumulation: f16(w) f16(x) exactly + fp4(w residual) fp4(f16 x) + fp4(f16 w) fp4(x residual), the
    fp4 operands block-scaled per (pixel, 16 channels) / per (output channel, tap, 16 input channels) as include/wsu.h (K1q) says."""
    xp = F.pad(x, (1, 1, 1, 1), mode="reflect")
    hi, ch, cr, e = planar_q_parts(xp)
    sc = torch.exp2(e)[..., None]
    xh, xc4, xr4 = _blocks_to_nchw(hi.float()), _blocks_to_nchw(fp4_values(ch) * sc), _blocks_to_nchw(fp4_values(cr) * sc / 2048.0)
    # weights: blocks of 16 input channels per (co, tap): move the input channel to the end
    wt = w.permute(0, 2, 3, 1).contiguous()                                    # (co, 3, 3, ci)
    co, _, _, ci = wt.shape
    whi, wch, wcr, we = planar_q_parts(wt.reshape(co * 9, ci, 1, 1))           # blocks along "channels" = ci
    wsc = torch.exp2(we)[..., None]
    back = lambda t: _blocks_to_nchw(t).reshape(co, 3, 3, ci).permute(0, 3, 1, 2)
    wh, wc4, wr4 = back(whi.float()), back(fp4_values(wch) * wsc), back(fp4_values(wcr) * wsc / 2048.0)
    y = F.conv2d(xh.double(), wh.double(), b.double()) + F.conv2d(xc4.double(), wr4.double()) + F.conv2d(xr4.double(), wc4.double())
    return y.float()


def _check_q_tensor(tq, ta, what):
    """A planar Q tensor against the e4m3-residual tensor `ta` the same kernel wrote from the same fp32 values: the f16 planes are the same
    bytes; block exponents and the f16 parts' fp4 nibbles are functions of the f16 planes -- exact; the residual nibbles come from the kernel's
    exact fp32 residual, `ta` carries it rounded to e4m3 -- a few land on the neighbouring grid point."""
    from gpu_util import fp4_codes, q_block_exp
    val, hi, ch, cr, e = planar_q_decode(tq, parts=True)
    raw = ta.detach().contiguous().view(torch.uint8)                                        # (n, chunk, 3, h, w, 16)
    n, nch, _, h, w, _ = raw.shape
    hi_a = torch.stack([raw[:, :, 0], raw[:, :, 1]], dim=-2).contiguous().view(torch.float16).reshape(n, nch, h, w, 16)
    res_a = raw[:, :, 2].contiguous().view(torch.float8_e4m3fn).float() / 4096.0
    assert torch.equal(hi.view(torch.int16), hi_a.view(torch.int16)), what + ": f16 planes"
    e_a = q_block_exp(hi_a.float().abs().amax(dim=-1))
    assert torch.equal(e, e_a), what + ": scale bytes"
    sc = torch.exp2(e_a)[..., None]
    assert torch.equal(fp4_values(ch), fp4_values(fp4_codes(hi_a.float() / sc))), what + ": fp4 nibbles of the f16 parts"
    want = fp4_values(fp4_codes(res_a * 2048.0 / sc))
    d = (fp4_values(cr) - want).abs()
    step = torch.where(want.abs() >= 4, 2.0, torch.where(want.abs() >= 2, 1.0, 0.5))
    assert bool((d <= step).all()), what + ": a residual nibble is more than one grid step off"
    assert float((d > 0).float().mean()) < 0.08, (what, float((d > 0).float().mean()))        # measured 1-5 %: the e4m3 rounding of the comparison value moves it across an fp4 rounding boundary
    return val


def _q_same(a, b) -> bool:
    """two planar Q tensors hold the same bytes wherever the format defines them (the scale plane's padding beyond the image is never written)"""
    from gpu_util import _q_sblock_index
    hw = a.h * a.w
    sidx = (48 * hw + _q_sblock_index(a.h, a.w, a.data.device)).reshape(-1)
    return (a.n, a.c, a.h, a.w) == (b.n, b.c, b.h, b.w) and torch.equal(a.data[:, :, :48 * hw], b.data[:, :, :48 * hw]) and torch.equal(a.data[:, :, sidx], b.data[:, :, sidx])


def test_producers_write_the_q_format():
    """Every producer of planar Q tensors, run once with y_format = PLANAR_A and once with PLANAR_Q on the same inputs: the Q tensor holds the
    f16 planes bit for bit, the block exponents and fp4 nibbles the format defines for those values (ragged sizes: partial scale-byte tiles)."""
    from ws_unet_amd import ops
    g = torch.Generator().manual_seed(3)
    # first layer
    x = torch.rand((2, 1, 40, 72), generator=g).to(DEV)
    w1, b1 = (torch.randn((64, 1, 3, 3), generator=g) * 0.5).to(DEV), (torch.randn(64, generator=g) * 0.1).to(DEV)
    ya = ops.conv3x3_first_pl(x, w1, b1)
    yq = ops.conv3x3_first_pl(x, w1, b1, y_format=ops.PLANAR_Q)
    assert isinstance(yq, ops.PlanarQ) and (yq.n, yq.c, yq.h, yq.w) == (2, 64, 40, 72)
    _check_q_tensor(yq, ya, "first_pl")
    # transposed conv (input always in the e4m3-residual format)
    xi = torch.relu(torch.randn((2, 64, 5, 37), generator=g))
    wt = (torch.randn((64, 64, 2, 2), generator=g) * 0.1).to(DEV)
    bt = (torch.randn(64, generator=g) * 0.1).to(DEV)
    wp = ops.pack_convt2x2(wt, ops.mode_id("f16f8"))
    ua = ops.convt2x2_pl(planar_encode(xi), wp, bt, 64)
    uq = ops.convt2x2_pl(planar_encode(xi), wp, bt, 64, y_format=ops.PLANAR_Q)
    _check_q_tensor(uq, ua, "convt2x2_pl")
    # the conv itself: plain, pooled
    xc = torch.relu(torch.randn((2, 32, 40, 72), generator=g))
    wc = (torch.randn((128, 32, 3, 3), generator=g) * (2.0 / (9 * 32)) ** 0.5).to(DEV)
    bc = (torch.randn(128, generator=g) * 0.1).to(DEV)
    wq = ops.pack_conv3x3_f4(wc)
    xq = planar_q_encode(xc)
    ya, pa = ops.conv3x3_q(xq, None, wq, bc, 128, pool=True, y_format=ops.PLANAR_A)
    yq, pq = ops.conv3x3_q(xq, None, wq, bc, 128, pool=True, y_format=ops.PLANAR_Q)
    _check_q_tensor(yq, ya, "conv3x3_q y")
    _check_q_tensor(pq, pa, "conv3x3_q y_pool")
    assert float((planar_decode(pa) - F.max_pool2d(planar_decode(ya), 2)).abs().max()) == 0.0
    y1 = ops.conv3x3_q(xq, None, wq, bc, 128, y_format=ops.PLANAR_Q)                     # no pool: another instantiation, same bytes
    assert _q_same(y1, yq)


def test_q_encode_decode_roundtrip_helpers():
    """The test-side encoder against its decoder (no kernel): values come back to within the format's residual precision."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn((1, 32, 19, 45), generator=g) * torch.exp2(torch.randint(-4, 5, (1, 32, 1, 1), generator=g).float())
    back = planar_q_decode(planar_q_encode(x))
    assert float((back - x).abs().max()) <= 2.0 ** -11 * float(x.abs().max())


@pytest.mark.parametrize("n,h,w,c1,c2,cout,pool", [
    (1, 16, 32, 64, 0, 64, False),            # one tile, 4 chunk steps
    (2, 40, 72, 64, 0, 128, False),           # partial tiles, 2 output blocks
    (2, 32, 64, 64, 64, 64, True),            # fused concat + pool
    (1, 96, 160, 128, 0, 64, False),          # 30 tiles x 8 steps
    (5, 128, 128, 32, 0, 64, False),          # 320 tiles of 2 steps: workgroups walk two tiles, the three-slot input ring wraps across tiles
    (1, 2, 2, 16, 0, 64, False),              # one chunk per tile (J = 1)
    (3, 18, 34, 48, 16, 64, False),           # odd chunk counts, sizes just past a tile
])
def test_conv3x3_q_matches_emulation(n, h, w, c1, c2, cout, pool):
    """f16 products on the f16 pipe + both cross terms as block-scaled fp4 -- equal to the CPU emulation of exactly that arithmetic up to accumulation
    order and the store encoding; and within 5e-4 of the exact convolution relative to the output's scale (VERDICT r03 next #2c: the measured
    level is 1.5e-4 .. 3e-4 on these operands -- zero-mean weights, ReLU inputs with per-channel scales 2^-3 .. 2^3; the bound per product is
    2^-13 relative to its block maxima -- 2e-3 was 20x the e4m3 band and would have passed a 10x regression)."""
    from ws_unet_amd import ops
    cin = c1 + c2
    g = torch.Generator().manual_seed(7)
    x = torch.relu(torch.randn((n, cin, h, w), generator=g)) * torch.exp2(torch.randint(-3, 4, (n, cin, 1, 1), generator=g).float())
    wgt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    ref = torch.relu(_conv3x3_q_ref(x, wgt, b))
    exact = torch.relu(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect").double(), wgt.double(), b.double())).float()
    wp = ops.pack_conv3x3_f4(wgt.to(DEV))
    x1 = planar_q_encode(x[:, :c1])
    x2 = planar_q_encode(x[:, c1:]) if c2 else None
    out = ops.conv3x3_q(x1, x2, wp, b.to(DEV), cout, pool=pool, y_format=ops.PLANAR_A)
    torch.cuda.synchronize()
    y = planar_decode(out[0] if pool else out)
    scale = float(exact.abs().max())
    assert float((y - ref).abs().max()) < 3e-5 * scale, float((y - ref).abs().max()) / scale
    assert float((y - exact).abs().max()) < 5e-4 * scale, float((y - exact).abs().max()) / scale
    if pool:
        assert float((planar_decode(out[1]) - F.max_pool2d(y, 2)).abs().max()) == 0.0


def test_conv3x3_q_variants_and_repeatability():
    """The other instantiations against the same emulation -- fused head (1 and 3 planes), small grids (half-block work items), no ReLU, pooled
    output alone -- and launch-to-launch repeatability of the three-slot / two-slot rings (a missing wait on a DMA piece shows as a difference
    between launches)."""
    from ws_unet_amd import ops
    g = torch.Generator().manual_seed(11)
    for hc in (1, 3):
        n, h, w, cin = 2, 24, 40, 64
        x = torch.relu(torch.randn((n, cin, h, w), generator=g))
        wgt = torch.randn((64, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn(64, generator=g) * 0.1
        hw_, hb = torch.randn((hc, 64, 1, 1), generator=g) * 0.2, torch.randn(hc, generator=g) * 0.1
        ref = torch.sigmoid(F.conv2d(torch.relu(_conv3x3_q_ref(x, wgt, b)), hw_, hb))
        xq, wp4 = planar_q_encode(x), ops.pack_conv3x3_f4(wgt.to(DEV))
        out = ops.conv3x3_q(xq, None, wp4, b.to(DEV), 64, head_w=hw_.to(DEV), head_b=hb.to(DEV), want_y=False)
        torch.cuda.synchronize()
        assert float((out.cpu() - ref).abs().max()) < 2e-5, (hc, float((out.cpu() - ref).abs().max()))
        out2, logit = ops.conv3x3_q(xq, None, wp4, b.to(DEV), 64, head_w=hw_.to(DEV), head_b=hb.to(DEV), want_y=False, want_logit=True)
        assert torch.equal(out2, out) and float((torch.sigmoid(logit) - out).abs().max()) < 1e-6
        # the head together with the stored activations: the same bits as each alone (a y beside the head is written in the e4m3-residual format)
        out_y = ops.conv3x3_q(xq, None, wp4, b.to(DEV), 64, head_w=hw_.to(DEV), head_b=hb.to(DEV), want_y=True)
        y_alone = ops.conv3x3_q(xq, None, wp4, b.to(DEV), 64, y_format=ops.PLANAR_A)
        assert torch.equal(out_y[0], out) and torch.equal(out_y[1].view(torch.int32), y_alone.view(torch.int32))
    # pooled output alone (no full-resolution store: an empty buffer descriptor drops it), ragged tile edges, both formats
    n, h, w, cin = 3, 40, 72, 64
    xe = planar_q_encode(torch.relu(torch.randn((n, cin, h, w), generator=g)))
    wp4 = ops.pack_conv3x3_f4((torch.randn((64, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(DEV))
    bz = (torch.randn(64, generator=g) * 0.1).to(DEV)
    for fmt in (ops.PLANAR_A, ops.PLANAR_Q):
        y_full, yp_full = ops.conv3x3_q(xe, None, wp4, bz, 64, pool=True, y_format=fmt)
        y_none, yp_only = ops.conv3x3_q(xe, None, wp4, bz, 64, pool=True, want_y=False, y_format=fmt)
        same = _q_same if fmt == ops.PLANAR_Q else (lambda u, v: torch.equal(u.view(torch.int32), v.view(torch.int32)))
        assert y_none is None and same(yp_only, yp_full)
    # small grid: 2 x (32 x 32) x 128 channels = 16 tiles -> half-block work items (kernel variant MSPLIT); and no ReLU; both formats
    n, h, w, cin, cout = 2, 32, 32, 128, 128
    x = torch.randn((n, cin, h, w), generator=g)
    wgt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    ref = _conv3x3_q_ref(x, wgt, b)
    wp = ops.pack_conv3x3_f4(wgt.to(DEV))
    ya = ops.conv3x3_q(planar_q_encode(x), None, wp, b.to(DEV), cout, relu=False, y_format=ops.PLANAR_A)
    assert float((planar_decode(ya) - ref).abs().max()) < 3e-5 * float(ref.abs().max())
    yq = ops.conv3x3_q(planar_q_encode(x), None, wp, b.to(DEV), cout, relu=False)
    _check_q_tensor(yq, ya, "msplit Q output")
    # repeatability at a size where every workgroup walks several tiles of 4 and of 16 steps
    for (n, s, cin, cout, pool) in ((8, 256, 64, 64, True), (4, 128, 256, 128, False)):
        xe = planar_q_encode(torch.relu(torch.randn((n, cin, s, s), generator=g)))
        wp = ops.pack_conv3x3_f4((torch.randn((cout, cin, 3, 3), generator=g) * 0.05).to(DEV))
        bz = torch.zeros(cout, device=DEV)
        first = ops.conv3x3_q(xe, None, wp, bz, cout, pool=pool)
        first = [ops.PlanarQ(t.data.clone(), t.n, t.c, t.h, t.w) for t in (first if pool else (first,))]
        for _ in range(10):
            again = ops.conv3x3_q(xe, None, wp, bz, cout, pool=pool)
            again = again if pool else (again,)
            for a_, f_ in zip(again, first):
                assert _q_same(a_, f_)


def test_conv3x3_q_argument_errors():
    """The entry points refuse what they do not implement, with a message (no silent fallback)."""
    from ws_unet_amd import ops, _lib
    with pytest.raises(_lib.WsuError, match="fp4 packing"):
        ops.pack_conv3x3_f4(torch.zeros((64, 8, 3, 3), device=DEV))                 # cin not a multiple of 16
    with pytest.raises(_lib.WsuError, match="fp4 packing"):
        ops.pack_conv3x3_f4(torch.zeros((32, 16, 3, 3), device=DEV))                # cout not a multiple of 64
    x = planar_q_encode(torch.zeros((1, 16, 8, 8)))
    wp = ops.pack_conv3x3_f4(torch.zeros((64, 16, 3, 3), device=DEV))
    with pytest.raises(_lib.WsuError, match="wsu_conv3x3_q_fwd"):                    # round 3's x_residual = 2 of the e4m3 kernel is gone, loudly
        ops.conv3x3_pl(planar_encode(torch.zeros((1, 16, 8, 8))), None, wp, None, 64, x_residual=2)
    with pytest.raises(_lib.WsuError, match="y_format"):
        ops.conv3x3_q(x, None, wp, None, 64, y_format=7)
    with pytest.raises(AssertionError, match="PlanarQ"):
        ops.conv3x3_q(planar_encode(torch.zeros((1, 16, 8, 8))), None, wp, None, 64)        # an e4m3-residual tensor is not a Q tensor
    with pytest.raises(AssertionError, match="pack_conv3x3_f4"):
        ops.conv3x3_q(x, None, ops.pack_conv3x3(torch.zeros((64, 16, 3, 3), device=DEV), ops.mode_id("f16f8")), None, 64)
    with pytest.raises(_lib.WsuError, match="fused pool needs even"):
        ops.conv3x3_q(planar_q_encode(torch.zeros((1, 16, 7, 8))), None, wp, None, 64, pool=True)
    with pytest.raises(_lib.WsuError, match="relu_mask_out"):
        ops.conv3x3_first_pl(torch.zeros((1, 1, 8, 8), device=DEV), torch.zeros((64, 1, 3, 3), device=DEV), None, want_mask=True, y_format=ops.PLANAR_Q)


def test_conv3x3_q_random_shapes():
    """A dozen randomly drawn problem shapes (fixed seed): image sizes that are not multiples of the 16 x 32 tile, 1-6 chunks per source, fused
    concat, pool on even sizes, 1-3 output blocks, negative inputs -- against the CPU emulation of the arithmetic."""
    from ws_unet_amd import ops
    rng = np.random.default_rng(20261005)
    g = torch.Generator().manual_seed(5)
    for _ in range(12):
        n = int(rng.integers(1, 4)); h = int(rng.integers(2, 70)); w = int(rng.integers(2, 100))
        c1 = 16 * int(rng.integers(1, 7)); c2 = 16 * int(rng.integers(0, 4)); cout = 64 * int(rng.integers(1, 4))
        pool = bool(rng.integers(0, 2)) and h % 2 == 0 and w % 2 == 0
        relu = bool(rng.integers(0, 2))
        x = torch.randn((n, c1 + c2, h, w), generator=g) * torch.exp2(torch.randint(-2, 3, (n, c1 + c2, 1, 1), generator=g).float())
        wgt = torch.randn((cout, c1 + c2, 3, 3), generator=g) * (2.0 / (9 * (c1 + c2))) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        ref = _conv3x3_q_ref(x, wgt, b)
        ref = torch.relu(ref) if relu else ref
        out = ops.conv3x3_q(planar_q_encode(x[:, :c1]), planar_q_encode(x[:, c1:]) if c2 else None, ops.pack_conv3x3_f4(wgt.to(DEV)), b.to(DEV), cout,
                            relu=relu, pool=pool, y_format=ops.PLANAR_A)
        y = planar_decode(out[0] if pool else out)
        scale = float(ref.abs().max())
        assert float((y - ref).abs().max()) < 3e-5 * scale, ((n, h, w, c1, c2, cout, pool, relu), float((y - ref).abs().max()) / scale)
        if pool:
            assert float((planar_decode(out[1]) - F.max_pool2d(y, 2)).abs().max()) == 0.0


def test_four_matrix_waves_give_the_same_bits():
    """WSU_Q_ROWS=4 (four matrix waves, one per SIMD, four tile rows each, explicit fragment pipeline: the experiment of VERDICT r03 next #3a,
    measured slower and kept as a switch) against the default (eight matrix waves x two rows): the same accumulation order per output tile, so
    the same bits -- and the organisation passes the emulation tests."""
    env = dict(os.environ, WSU_Q_ROWS="4")
    r = subprocess.run([sys.executable, "-m", "pytest", str(HERE / "test_gpu_q.py"), "-q", "-x", "-k", "matches_emulation or variants or producers"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from gpu_util import planar_q_encode\nfrom ws_unet_amd import ops\n"
            "g = torch.Generator().manual_seed(2)\n"
            "x = planar_q_encode(torch.relu(torch.randn((3, 128, 64, 128), generator=g)))\n"
            "w = (torch.randn((128, 128, 3, 3), generator=g) * 0.04).cuda(); b = torch.zeros(128, device='cuda')\n"
            "y, yp = ops.conv3x3_q(x, None, ops.pack_conv3x3_f4(w), b, 128, pool=True)\n"
            "torch.cuda.synchronize(); torch.save((y.data.cpu(), yp.data.cpu()), sys.argv[1])\n") % (str(HERE.parent), str(HERE))      # (64 x 128, pooled 32 x 64: whole scale-byte tiles, no padding bytes)
    outs = []
    for rows in ("4", "2"):
        out = str(HERE / f".qrows_{rows}.pt")
        rr = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, WSU_Q_ROWS=rows), capture_output=True, text=True, timeout=300)
        assert rr.returncode == 0, rr.stderr[-2000:]
        outs.append(torch.load(out, weights_only=True)); os.remove(out)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("n,h,w,cout", [(2, 64, 96, 64), (1, 16, 32, 128), (3, 50, 70, 64), (1, 512, 512, 64)])
def test_fused_first_q_is_bitwise_the_two_kernel_path(n, h, w, cout):
    """e11 + e12 + pool in one launch (kernel variant F1: the loader waves compute e11's channels into the LDS input slots) against
    conv3x3_first_pl(y_format Q) -> conv3x3_q(pool): the same fp32 FMA order, the same planar-Q encoding -- the same bytes; the range flag covers the
    computed (never stored) xe11 values."""
    from ws_unet_amd import ops
    g = torch.Generator().manual_seed(31)
    x = torch.rand((n, 1, h, w), generator=g).to(DEV)
    w1, b1 = (torch.randn((64, 1, 3, 3), generator=g) * 0.5).to(DEV), (torch.randn(64, generator=g) * 0.1).to(DEV)
    w2 = (torch.randn((cout, 64, 3, 3), generator=g) * (2.0 / (9 * 64)) ** 0.5).to(DEV)
    b2 = (torch.randn(cout, generator=g) * 0.1).to(DEV)
    wp = ops.pack_conv3x3_f4(w2)
    xe11 = ops.conv3x3_first_pl(x, w1, b1, y_format=ops.PLANAR_Q)
    y0, p0 = ops.conv3x3_q(xe11, None, wp, b2, cout, pool=True)
    y1, p1 = ops.conv3x3_q_fused_first(x, w1, b1, wp, b2, cout)
    assert _q_same(y1, y0) and _q_same(p1, p0)
    for _ in range(5):
        y2, p2 = ops.conv3x3_q_fused_first(x, w1, b1, wp, b2, cout)
        assert _q_same(y2, y0) and _q_same(p2, p0)
    rf = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.conv3x3_q_fused_first(x, w1, b1, wp, b2, cout, range_flag=rf)
    assert int(rf.item()) == 0
    ops.conv3x3_q_fused_first(x, w1 * 4000.0, b1, wp, b2 * 0, cout, range_flag=rf)        # xe11 beyond +-448: only the loaders see it
    assert int(rf.item()) == 1


def test_whole_net_with_and_without_the_fused_first_layer():
    from gpu_util import gpu_model, images01
    _, x = images01(2, 96, 160, seed=3)
    m = gpu_model(2, "he", "f16f4p")
    with torch.no_grad():
        m.fuse_first_q = True
        y1 = m(x.to(DEV)).cpu()
        m.fuse_first_q = False
        y0 = m(x.to(DEV)).cpu()
    assert torch.equal(y0, y1)
