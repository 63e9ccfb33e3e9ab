"""Round 4: a decoder block's transposed conv + concat + first 3x3 conv in one launch (csrc/conv3x3_qu.hip, include/wsu.h K1u) --
relu(conv3x3_reflect(cat[conv_transpose2x2_s2(x_low), skip])), unet.py:171-173 / 177-179 / 183-185 -- against (a) the exact composition of the two
reference ops in fp64 and (b) a CPU restatement of the kernel's own arithmetic (parity-class 2x2-tap weights combined in fp32, f16 products + block-scaled
fp4 cross terms on planar Q operands).  Every call goes through the C ABI."""
import pytest
import torch
import torch.nn.functional as F

from gpu_util import DEV, fp4_values, planar_q_decode, planar_q_encode, planar_q_parts
from test_gpu_q import _blocks_to_nchw

pytestmark = pytest.mark.gpu


def _combined_weights(w3, wt, cup):
    """Wc[co][c][py][px][dy][dx] = sum_ci sum_{(ky,kx) -> (dy,dx)} w3[co][ci][ky][kx] wt[c][ci][sy][sx] in fp64 (include/wsu.h K1u)."""
    cout, cl = w3.shape[0], wt.shape[0]
    wc = torch.zeros((cout, cl, 2, 2, 2, 2), dtype=torch.float64)
    for py in range(2):
        for ky in range(3):
            r = py + ky - 1                                         # row of the upsampled tensor relative to 2 i
            dy, sy = r // 2 + 1 - py, r % 2
            for px in range(2):
                for kx in range(3):
                    c = px + kx - 1
                    dx, sx = c // 2 + 1 - px, c % 2
                    wc[:, :, py, px, dy, dx] += torch.einsum("oi,ci->oc", w3[:, :cup, ky, kx].double(), wt[:, :, sy, sx].double())
    return wc


def _q_conv_terms(xp, w):
    """sum of the three product families of the fp4-cross-term arithmetic for a VALID conv of the padded input xp with w (any kernel size), fp64:
    f16(w) f16(x) + fp4(w residual) fp4(f16 x) + fp4(f16 w) fp4(x residual); blocks = 16 channels per pixel / per (co, tap)."""
    hi, ch, cr, e = planar_q_parts(xp)
    sc = torch.exp2(e)[..., None]
    xh, xc4, xr4 = _blocks_to_nchw(hi.float()), _blocks_to_nchw(fp4_values(ch) * sc), _blocks_to_nchw(fp4_values(cr) * sc / 2048.0)
    co, ci, kh, kw = w.shape
    wt = w.permute(0, 2, 3, 1).contiguous()
    whi, wch, wcr, we = planar_q_parts(wt.reshape(co * kh * kw, ci, 1, 1))
    wsc = torch.exp2(we)[..., None]
    back = lambda t: _blocks_to_nchw(t).reshape(co, kh, kw, ci).permute(0, 3, 1, 2)
    wh, wc4, wr4 = back(whi.float()), back(fp4_values(wch) * wsc), back(fp4_values(wcr) * wsc / 2048.0)
    return F.conv2d(xh.double(), wh.double()) + F.conv2d(xc4.double(), wr4.double()) + F.conv2d(xr4.double(), wc4.double())


def _up_q_ref(xl, xs, w3, wc, bias, cup):
    """the kernel's arithmetic on the CPU: ordinary 3x3 terms on the skip half + per parity class a 2x2-tap conv on the clamp-padded low tensor"""
    _, _, hl, wl = xl.shape
    y = _q_conv_terms(F.pad(xs, (1, 1, 1, 1), mode="reflect"), w3[:, cup:])
    xlp = F.pad(xl, (1, 1, 1, 1), mode="replicate")
    for py in range(2):
        for px in range(2):
            t = _q_conv_terms(xlp, wc[:, :, py, px].float())                 # (n, co, hl + 1, wl + 1): output (i, j) reads padded rows i, i + 1
            y[:, :, py::2, px::2] += t[:, :, py:py + hl, px:px + wl]
    return (y + bias.double()[None, :, None, None]).float()


def _q_roundtrip(v):
    """fp32 NCHW -> what a planar Q tensor keeps of it: f16 part + fp4 residual * 2^(E - 11)"""
    hi, _, cr, e = planar_q_parts(v)
    return _blocks_to_nchw(hi.float() + fp4_values(cr) * torch.exp2(e - 11)[..., None])


def _case(n, hl, wl, cl, cup, c2, cout, seed=11):
    g = torch.Generator().manual_seed(seed)
    xl = torch.relu(torch.randn((n, cl, hl, wl), generator=g)) * torch.exp2(torch.randint(-3, 4, (n, cl, 1, 1), generator=g).float())
    xs = torch.relu(torch.randn((n, c2, 2 * hl, 2 * wl), generator=g)) * torch.exp2(torch.randint(-3, 4, (n, c2, 1, 1), generator=g).float())
    wt = torch.randn((cl, cup, 2, 2), generator=g) * (1.0 / cl) ** 0.5
    bt = torch.randn(cup, generator=g) * 0.1
    w3 = torch.randn((cout, cup + c2, 3, 3), generator=g) * (2.0 / (9 * (cup + c2))) ** 0.5
    b3 = torch.randn(cout, generator=g) * 0.1
    return xl, xs, wt, bt, w3, b3


@pytest.mark.parametrize("n,hl,wl,cl,cup,c2,cout", [
    (1, 8, 16, 32, 16, 16, 64),               # one tile, one skip chunk, two low chunks
    (2, 16, 32, 128, 64, 64, 64),             # d41's channels: 4 skip steps + 16 low half-steps per tile
    (1, 24, 40, 64, 32, 32, 128),             # two output blocks, tiles past the image in both directions (48 x 80)
    (3, 5, 7, 16, 16, 48, 64),                # 10 x 14 pixels: a single ragged tile, more skip than low chunks
    (1, 64, 64, 32, 16, 32, 64),              # 32 tiles
    (8, 64, 128, 32, 16, 16, 128),            # 1024 work items on 256 CUs: every workgroup walks four tiles, the ring and the loaders' cursor cross tile borders
])
def test_conv3x3_up_q_matches_the_composition_and_its_emulation(n, hl, wl, cl, cup, c2, cout):
    from ws_unet_amd import ops
    xl, xs, wt, bt, w3, b3 = _case(n, hl, wl, cl, cup, c2, cout)
    # (a) the reference's two ops, fp64
    xu = F.conv_transpose2d(xl.double(), wt.double(), bt.double(), stride=2)
    exact = torch.relu(F.conv2d(F.pad(torch.cat([xu, xs.double()], 1), (1, 1, 1, 1), mode="reflect"), w3.double(), b3.double())).float()
    w_skip, w_low, bias, dense = ops.pack_conv3x3_up(w3.to(DEV), wt.to(DEV), bt.to(DEV), b3.to(DEV), want_dense=True)
    # the combined weights and bias against fp64
    wc64 = _combined_weights(w3, wt, cup)
    assert float((dense.cpu().double() - wc64).abs().max()) < 2e-6 * float(wc64.abs().max())
    bias64 = b3.double() + torch.einsum("oikl,i->o", w3[:, :cup].double(), bt.double())
    assert float((bias.cpu().double() - bias64).abs().max()) < 1e-5
    y = ops.conv3x3_up_q(planar_q_encode(xl), planar_q_encode(xs), w_skip, w_low, bias, cout)
    torch.cuda.synchronize()
    got = planar_q_decode(y)
    scale = float(exact.abs().max())
    # (b) the kernel's arithmetic restated (the device's fp32 combined weights, so that the f16 / fp4 splits are the same numbers)
    ref = torch.relu(_up_q_ref(xl, xs, w3, dense.cpu(), bias.cpu(), cup))
    # a planar Q tensor keeps f16 + a 2.5-bit residual (up to 2^-13 of the pixel's block maximum away from the value): compare with the emulation
    # THROUGH the same encoding -- equal up to the accumulation order except where that moves a value across a rounding boundary of the encoding
    d = (got - _q_roundtrip(ref)).abs()
    assert float((d > 3e-5 * scale).float().mean()) < 0.02, float((d > 3e-5 * scale).float().mean())
    assert float(d.max()) < 2.5e-4 * scale and float((got - ref).abs().max()) < 1.6e-4 * scale, (float(d.max()) / scale, float((got - ref).abs().max()) / scale)
    assert float((got - exact).abs().max()) < 5e-4 * scale, float((got - exact).abs().max()) / scale


def test_conv3x3_up_q_equals_the_two_kernel_path_within_the_format():
    """the fused launch against convt2x2_pl -> conv3x3_q on the same operands: both approximate the same composition; the fused one is not further from
    the exact result than the two-kernel path (no rounding of the upsampled tensor to storage)"""
    from ws_unet_amd import ops
    from gpu_util import planar_encode
    n, hl, wl, cl, cup, c2, cout = 2, 24, 48, 128, 64, 64, 64
    xl, xs, wt, bt, w3, b3 = _case(n, hl, wl, cl, cup, c2, cout, seed=5)
    xu = F.conv_transpose2d(xl.double(), wt.double(), bt.double(), stride=2)
    exact = torch.relu(F.conv2d(F.pad(torch.cat([xu, xs.double()], 1), (1, 1, 1, 1), mode="reflect"), w3.double(), b3.double())).float()
    w_skip, w_low, bias = ops.pack_conv3x3_up(w3.to(DEV), wt.to(DEV), bt.to(DEV), b3.to(DEV))
    fused = planar_q_decode(ops.conv3x3_up_q(planar_q_encode(xl), planar_q_encode(xs), w_skip, w_low, bias, cout))
    xuq = ops.convt2x2_pl(planar_encode(xl), ops.pack_convt2x2(wt.to(DEV), ops.mode_id("f16f8")), bt.to(DEV), cup, y_format=ops.PLANAR_Q)
    two = planar_q_decode(ops.conv3x3_q(xuq, planar_q_encode(xs), ops.pack_conv3x3_f4(w3.to(DEV)), b3.to(DEV), cout))
    e_f, e_t = float((fused - exact).abs().mean()), float((two - exact).abs().mean())
    assert e_f <= 1.1 * e_t, (e_f, e_t)
    assert float((fused - two).abs().max()) < 6e-4 * float(exact.abs().max())


def test_conv3x3_up_q_repeatable_and_argument_errors():
    from ws_unet_amd import ops, _lib
    n, hl, wl, cl, cup, c2, cout = 2, 40, 40, 64, 32, 32, 64
    xl, xs, wt, bt, w3, b3 = _case(n, hl, wl, cl, cup, c2, cout, seed=3)
    w_skip, w_low, bias = ops.pack_conv3x3_up(w3.to(DEV), wt.to(DEV), bt.to(DEV), b3.to(DEV))
    ql, qs = planar_q_encode(xl), planar_q_encode(xs)
    first = ops.conv3x3_up_q(ql, qs, w_skip, w_low, bias, cout).data.clone()
    hw = 4 * hl * wl
    for _ in range(20):                                          # the ring, the waits and the barriers leave no launch-to-launch difference
        again = ops.conv3x3_up_q(ql, qs, w_skip, w_low, bias, cout).data
        assert torch.equal(again[:, :, :48 * hw], first[:, :, :48 * hw])
    # the range flag fires on a stored value beyond +-448
    rf = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.conv3x3_up_q(ql, qs, w_skip, w_low, bias, cout, range_flag=rf)
    assert int(rf.item()) == 0
    ops.conv3x3_up_q(ql, qs, w_skip, w_low, bias + 1000.0, cout, range_flag=rf)
    assert int(rf.item()) == 1
    lib = _lib.load()
    assert lib.wsu_conv3x3_up_packed_bytes(24, 64) == 0 and lib.wsu_conv3x3_up_packed_bytes(32, 96) == 0
    args = lambda h, w, cl_, c2_, co: (ql.data_ptr(), qs.data_ptr(), w_skip.data_ptr(), w_low.data_ptr(), bias.data_ptr(), first.data_ptr(), n, h, w, cl_, c2_, co, 1, None, None)
    ERR_ARG = -1                                                 # include/wsu.h WSU_ERR_ARG
    assert lib.wsu_conv3x3_up_q_fwd(*args(81, 80, cl, c2, cout)) == ERR_ARG and b"even" in lib.wsu_last_error()
    assert lib.wsu_conv3x3_up_q_fwd(*args(80, 80, 24, c2, cout)) == ERR_ARG
    assert lib.wsu_conv3x3_up_q_fwd(*args(80, 80, cl, 0, cout)) == ERR_ARG
    assert lib.wsu_conv3x3_up_q_fwd(*args(80, 80, cl, c2, 1024)) == ERR_ARG
    with pytest.raises(_lib.WsuError, match="fused upsample packing"):
        ops.pack_conv3x3_up(w3[:, :cup + 8].contiguous().to(DEV), wt.to(DEV), None, None)
    with pytest.raises(AssertionError, match="half the skip"):
        ops.conv3x3_up_q(qs, qs, w_skip, w_low, bias, cout)


@pytest.mark.parametrize("nsteps,size", [(2, 96), (4, 64)])
def test_whole_net_fused_decoder_entries_against_the_two_kernel_path_and_the_oracle(nsteps, size):
    """UNet in the default mode: every decoder block's entry as one launch (default) against convt2x2_pl -> conv3x3_q (fuse_up_planar = False, the
    round's earlier path) and against the fp32 CPU oracle -- the fused path is at least as close to the oracle."""
    from gpu_util import gpu_model, images01, oracle_forward
    _, x = images01(2, size, size + 32, seed=21)
    ref = oracle_forward(x, nsteps, "he")
    m = gpu_model(nsteps, "he", "f16f4p")
    assert m.fuse_up_planar
    with torch.no_grad():
        y_f = m(x.to(DEV)).cpu()
        m.fuse_up_planar = False
        y_t = m(x.to(DEV)).cpu()
        m.fuse_up_planar = True
        y_f2 = m(x.to(DEV)).cpu()
    assert torch.equal(y_f, y_f2)
    e_f, e_t = float((y_f - ref).abs().mean()), float((y_t - ref).abs().mean())
    assert e_f < 1e-4 and e_f <= 1.15 * e_t, (e_f, e_t)
    assert float((y_f - y_t).abs().max()) < 6e-4


@pytest.mark.parametrize("n,h,w", [(1, 1024, 1024), (2, 384, 640), (1, 2048, 1536)])
def test_default_mode_at_the_pair_training_resolution_and_non_square(n, h, w):
    """BASELINE.json configs[4] runs 1024x1024 images: the default inference path (fused first layer, fused decoder entries: many tiles per
    workgroup, the ring wrapping across hundreds of tiles) against the fp32 CPU oracle at that size, and at a non-square size."""
    from gpu_util import gpu_model, images01, oracle_forward
    _, x = images01(n, h, w, seed=13)
    ref = oracle_forward(x, 2, "he")
    m = gpu_model(2, "he", None)
    with torch.no_grad():
        y = m(x.to(DEV)).cpu()
        y2 = m(x.to(DEV)).cpu()
    assert torch.equal(y, y2)
    assert float((y - ref).abs().mean()) <= 5e-5 and float((y - ref).abs().max()) <= 9e-4, (float((y - ref).abs().mean()), float((y - ref).abs().max()))
