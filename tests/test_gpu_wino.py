"""GPU tests of the Winograd F(2,3) forward conv (wsu_conv3x3_wino_fwd, mode bf16x3) against the oracle and the direct kernel."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import DEV, ACT_RTOL, from_nhwc, rand_act, to_nhwc
from ws_unet_amd import formula, ops
from oracle import np_ops, unet_ref

pytestmark = pytest.mark.gpu
MODE = "bf16x3"


def _close(got, ref, what, rtol=None):
    rtol = rtol or 2 * ACT_RTOL[MODE]                       # F(2,3): one extra add/sub level of fp32 re-association
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


# (N, H, W, C1, C2, Cout): ragged sizes, row/column counts not multiples of the 16x32 tile, concat, 2x2 minimum
SHAPES = [(2, 16, 32, 64, 0, 64), (1, 37, 70, 64, 0, 128), (1, 24, 40, 64, 64, 64), (1, 2, 2, 64, 0, 64),
          (1, 3, 5, 128, 0, 64), (1, 33, 31, 128, 128, 128)]


@pytest.mark.parametrize("shape", SHAPES)
def test_wino_conv_vs_oracle_and_direct(shape):
    n, h, w, c1, c2, cout = shape
    m = ops.mode_id(MODE)
    x1 = rand_act((n, c1, h, w), f"wn/x1/{shape}")
    x2 = rand_act((n, c2, h, w), f"wn/x2/{shape}") if c2 else None
    wt = torch.from_numpy(formula.formula_tensor(f"wn/w/{shape}", (cout, c1 + c2, 3, 3), (6.0 / (9 * (c1 + c2))) ** 0.5))
    b = torch.from_numpy(formula.formula_tensor(f"wn/b/{shape}", (cout,), 0.1))
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref = F.relu(unet_ref.conv3x3_reflect(xin, wt, b))
    a1, a2 = to_nhwc(x1, MODE), None if x2 is None else to_nhwc(x2, MODE)
    wpw = ops.pack_conv3x3_wino(wt.to(DEV))
    y = ops.conv3x3_wino(a1, a2, wpw, b.to(DEV), cout, relu=True)
    _close(from_nhwc(y), ref, f"wino {shape}")
    yd = ops.conv3x3(a1, a2, ops.pack_conv3x3(wt.to(DEV), m), b.to(DEV), cout, m, relu=True)
    _close(from_nhwc(y), from_nhwc(yd), f"wino vs direct {shape}")
    y2 = ops.conv3x3_wino(a1, a2, wpw, None, cout, relu=False)
    _close(from_nhwc(y2), unet_ref.conv3x3_reflect(xin, wt, None), f"wino linear {shape}")
    assert torch.equal(y, ops.conv3x3_wino(a1, a2, wpw, b.to(DEV), cout, relu=True))          # deterministic


def test_wino_fused_pool_and_head():
    n, h, w, c, cout = 2, 24, 40, 64, 64
    x = rand_act((n, c, h, w), "wn/pool/x")
    wt = torch.from_numpy(formula.formula_tensor("wn/pool/w", (cout, c, 3, 3), (6.0 / (9 * c)) ** 0.5))
    b = torch.from_numpy(formula.formula_tensor("wn/pool/b", (cout,), 0.5))
    wpw = ops.pack_conv3x3_wino(wt.to(DEV))
    y, yp, idx = ops.conv3x3_wino(to_nhwc(x, MODE), None, wpw, b.to(DEV), cout, pool=True, pool_idx=True)
    ref_pool, ref_arg = np_ops.maxpool2x2(from_nhwc(y).numpy())
    np.testing.assert_array_equal(from_nhwc(yp).numpy(), ref_pool)                         # bitwise: pool of the kernel's own output
    np.testing.assert_array_equal(idx.permute(0, 3, 1, 2).cpu().numpy(), ref_arg)
    # fused 1x1 head + sigmoid
    hw = torch.from_numpy(formula.formula_tensor("wn/head/w", (1, cout, 1, 1), 0.3))
    hb = torch.from_numpy(formula.formula_tensor("wn/head/b", (1,), 0.1))
    out, logit, yy = ops.conv3x3_wino(to_nhwc(x, MODE), None, wpw, b.to(DEV), cout, head_w=hw.to(DEV), head_b=hb.to(DEV),
                                      want_logit=True, want_y=True)
    assert torch.equal(yy, y)
    ref_logit = F.conv2d(from_nhwc(y), hw, hb)
    np.testing.assert_allclose(logit.cpu().numpy(), ref_logit.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(out.cpu().numpy(), torch.sigmoid(ref_logit).numpy(), rtol=0, atol=5e-6)
    out2 = ops.conv3x3_wino(to_nhwc(x, MODE), None, wpw, b.to(DEV), cout, head_w=hw.to(DEV), head_b=hb.to(DEV), want_y=False)
    assert torch.equal(out2, out)


def test_wino_argument_errors():
    x = torch.zeros((1, 8, 8, 64), device=DEV)
    with pytest.raises(Exception, match="cin"):
        ops.pack_conv3x3_wino(torch.zeros((64, 8, 3, 3), device=DEV))
    wpw = ops.pack_conv3x3_wino(torch.zeros((64, 64, 3, 3), device=DEV))
    with pytest.raises(Exception, match="bad shape"):
        ops.conv3x3_wino(torch.zeros((1, 1, 8, 64), device=DEV), None, wpw, None, 64)
    with pytest.raises(Exception, match="CPU tensor"):
        ops.conv3x3_wino(x.cpu(), None, wpw, None, 64)
