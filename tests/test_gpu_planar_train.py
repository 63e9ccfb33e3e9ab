"""Parity of the PLANAR training kernels (gradients in the F16F8P layout) against torch-CPU autograd of the reference's layers
(src/unet/model/unet.py:82-132,141-189 under autograd; oracle: torch.nn.functional on the CPU, fp32).

Tolerances: the f16f8 arithmetic keeps ~2^-15 per product and the stored gradients ~2^-16 per value, so a gradient tensor agrees with the
fp32 oracle to a relative L2 error of ~1e-4 (band 3e-4) -- the bands of tests/test_gpu_backward_large.py for the NHWC f16f8x kernels.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import DEV, GRAD_LO, planar_decode, planar_encode

pytestmark = pytest.mark.gpu

REL_L2 = 3e-4


def _ops():
    from ws_unet_amd import ops
    return ops


def rel_l2(got: torch.Tensor, ref: torch.Tensor) -> float:
    return float((got.double() - ref.double()).norm() / ref.double().norm().clamp_min(1e-30))


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _q(t, lo):
    """the values a planar tensor holds for t"""
    return planar_decode(planar_encode(t, lo), lo)


@pytest.mark.parametrize("n,h,w,cin,csplit,cout,masked", [
    (2, 16, 32, 64, 64, 64, False),
    (1, 40, 72, 64, 64, 64, True),          # partial tiles in both directions
    (2, 32, 64, 128, 64, 64, False),        # fused concat: two gradients
    (1, 24, 40, 128, 128, 32, True),
    (1, 3, 5, 64, 64, 64, True),            # rows 1 and h-2 coincide
    (2, 2, 2, 64, 64, 32, False),
])
@pytest.mark.parametrize("pad_zero", [True, False])
def test_conv3x3_pl_bwd_data(n, h, w, cin, csplit, cout, masked, pad_zero):
    ops = _ops()
    wgt = _rand((cout, cin, 3, 3), 1, (2.0 / (9 * cin)) ** 0.5)
    g = _q(_rand((n, cout, h, w), 2), GRAD_LO)
    act = torch.relu(_rand((n, cin, h, w), 3)) if masked else None
    x = torch.zeros((n, cin, h, w), requires_grad=True)
    xp = F.pad(x, (1, 1, 1, 1), mode="constant" if pad_zero else "reflect")
    F.conv2d(xp, wgt).backward(g)
    ref = x.grad.clone()
    if masked:
        ref = ref * (act > 0)
    wd = wgt.to(DEV)
    wp = ops.pack_conv3x3(wd, ops.MODE_F16F8, dgrad=True)
    wr = None if pad_zero else ops.pack_conv3x3_ring(wd)
    m1 = planar_encode(act[:, :csplit]) if masked else None
    m2 = planar_encode(act[:, csplit:]) if (masked and csplit < cin) else None
    dx1, dx2 = ops.conv3x3_pl_bwd_data(planar_encode(g, GRAD_LO), wp, wr, cin, csplit, m1, m2, pad_zero=pad_zero)
    torch.cuda.synchronize()
    got = planar_decode(dx1, GRAD_LO)
    if dx2 is not None:
        got = torch.cat([got, planar_decode(dx2, GRAD_LO)], dim=1)
    assert got.shape == ref.shape
    assert rel_l2(got, ref) < REL_L2, rel_l2(got, ref)
    assert float((got - ref).abs().max()) < 2e-3 * float(ref.abs().max())
    if masked:
        assert float(got[(act <= 0)].abs().max()) == 0.0


@pytest.mark.parametrize("n,h,w,c1,c2,cout", [
    (2, 16, 32, 64, 0, 64),
    (1, 37, 70, 64, 0, 128),            # partial tiles, several tiles per image column (rolling row window)
    (2, 24, 40, 64, 64, 64),            # fused concat
    (3, 10, 33, 128, 0, 64),
])
def test_conv3x3_pl_bwd_weight(n, h, w, c1, c2, cout):
    ops = _ops()
    cin = c1 + c2
    x = _q(torch.relu(_rand((n, cin, h, w), 5)), 4096.0)
    g = _q(_rand((n, cout, h, w), 6), GRAD_LO)
    wgt = torch.zeros((cout, cin, 3, 3), requires_grad=True)
    b = torch.zeros(cout, requires_grad=True)
    F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), wgt, b).backward(g)
    x1 = planar_encode(x[:, :c1])
    x2 = planar_encode(x[:, c1:]) if c2 else None
    dw, db = ops.conv3x3_pl_bwd_weight(planar_encode(g, GRAD_LO), x1, x2)
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu(), wgt.grad) < REL_L2, rel_l2(dw.cpu(), wgt.grad)
    assert rel_l2(db.cpu(), b.grad) < 2e-6, rel_l2(db.cpu(), b.grad)
    dw2, _ = ops.conv3x3_pl_bwd_weight(planar_encode(g, GRAD_LO), x1, x2)
    assert torch.equal(dw, dw2)                                        # fixed-order reduction


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 8, 32, 64, 64), (1, 13, 40, 128, 64), (2, 5, 7, 64, 128)])
def test_convt2x2_pl_bwd_weight(n, h, w, cin, cout):
    ops = _ops()
    x = _q(torch.relu(_rand((n, cin, h, w), 7)), 4096.0)
    dy = _q(_rand((n, cout, 2 * h, 2 * w), 8), GRAD_LO)
    wgt = torch.zeros((cin, cout, 2, 2), requires_grad=True)
    F.conv_transpose2d(x, wgt, stride=2).backward(dy)
    dw = ops.convt2x2_pl_bwd_weight(planar_encode(x), planar_encode(dy, GRAD_LO))
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu(), wgt.grad) < REL_L2, rel_l2(dw.cpu(), wgt.grad)
