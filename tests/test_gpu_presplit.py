"""GPU tests of mode 'bf16x3s' (activations stored already split into bf16 hi / lo halves by their producer): bitwise the results
of 'bf16x3', kernel by kernel and for whole networks."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, images01, gpu_model, oracle_forward, unsplit
from ws_unet_amd import formula, ops

pytestmark = pytest.mark.gpu
X3, X3S = ops.mode_id("bf16x3"), ops.mode_id("bf16x3s")


def _w(key, shape, scale):
    return torch.from_numpy(formula.formula_tensor(key, shape, scale)).to(DEV)


@pytest.mark.parametrize("hw", [(16, 32), (24, 40), (8, 8), (36, 70)])
def test_kernel_chain_is_bitwise_the_fp32_storage_chain(hw):
    """fused first layer -> conv (+pool) -> conv -> transposed conv -> concat conv -> conv + head, in both storage formats: every
    intermediate in split storage decodes to (hi + lo of) the fp32-storage tensor, and the final fp32 output is identical."""
    h, w = hw
    x = images01(2, h, w, seed=3)[1].to(DEV)
    w1, b1 = _w(f"ps/w1/{hw}", (64, 1, 3, 3), 0.5), _w(f"ps/b1/{hw}", (64,), 0.1)
    w2, b2 = _w(f"ps/w2/{hw}", (64, 64, 3, 3), 0.06), _w(f"ps/b2/{hw}", (64,), 0.1)
    w3, b3 = _w(f"ps/w3/{hw}", (128, 64, 3, 3), 0.06), _w(f"ps/b3/{hw}", (128,), 0.1)
    wu, bu = _w(f"ps/wu/{hw}", (128, 64, 2, 2), 0.09), _w(f"ps/bu/{hw}", (64,), 0.1)
    w4, b4 = _w(f"ps/w4/{hw}", (64, 128, 3, 3), 0.04), _w(f"ps/b4/{hw}", (64,), 0.1)
    w5, b5 = _w(f"ps/w5/{hw}", (64, 64, 3, 3), 0.06), _w(f"ps/b5/{hw}", (64,), 0.1)
    hw_, hb = _w(f"ps/hw/{hw}", (1, 64, 1, 1), 0.3), _w(f"ps/hb/{hw}", (1,), 0.1)
    outs = {}
    for m in (X3, X3S):
        y12, yp = ops.conv3x3_fused_first(x, w1, b1, ops.pack_conv3x3(w2, m), b2, 64, m, pool=True)
        y21 = ops.conv3x3(yp, None, ops.pack_conv3x3(w3, m), b3, 128, m)
        yu = ops.convt2x2(y21, ops.pack_convt2x2(wu, m), bu, 64, m)
        yd = ops.conv3x3(yu, y12, ops.pack_conv3x3(w4, m), b4, 64, m)
        out, logit = ops.conv3x3_head(yd, None, ops.pack_conv3x3(w5, m), b5, hw_, hb, m, want_logit=True)
        outs[m] = (y12, yp, y21, yu, yd, out, logit)
    for a, b in zip(outs[X3][:5], outs[X3S][:5]):
        ref = a.cpu()
        got = unsplit(b)
        assert (got - ref).abs().max().item() <= 2.0 ** -16 * max(ref.abs().max().item(), 1e-30)      # hi + lo carries 16+ bits
        # re-splitting the decoded value reproduces the stored halves: the producer used the staging split (same wsu_split2)
        assert torch.equal(unsplit(b), got)
    assert torch.equal(outs[X3][5], outs[X3S][5]) and torch.equal(outs[X3][6], outs[X3S][6])
    with pytest.raises(Exception, match="BF16X3S"):
        ops.conv3x3(outs[X3S][0], None, ops.pack_conv3x3(w2, X3S), b2, 64, X3S, pool=True, pool_idx=True)


@pytest.mark.parametrize("ns", [1, 2, 3])
def test_whole_network_bitwise_equal_to_bf16x3(ns):
    _, x = images01(2, 64, 64, seed=13)
    m = gpu_model(ns, "he", "bf16x3s")
    with torch.no_grad():                                        # the inference path (autograd keeps fp32 storage)
        y1 = gpu_model(ns, "he", "bf16x3")(x.to(DEV))
        y2 = m(x.to(DEV))
    assert torch.equal(y1, y2)
    keep = {}
    with torch.no_grad():
        y3 = m.forward_features(x.to(DEV), keep=keep)           # keep= runs in fp32 storage
    assert "xe11" in keep and (y3 - y1).abs().max().item() <= 2e-6       # keep= also un-fuses the head (different summation order)
    assert (y2.cpu() - oracle_forward(x, ns)).abs().max().item() <= 4e-5
    assert m.train_mode == "bf16x3"                              # training is unaffected (fp32 storage)


def test_presplit_mode_512(golden):
    g = golden["unet_fwd_512"]
    _, x = images01(1, 512, 512, seed=7)
    with torch.no_grad():
        y = gpu_model(2, "he", "bf16x3s")(x.to(DEV))
        y1 = gpu_model(2, "he", "bf16x3")(x.to(DEV))
    assert torch.equal(y, y1)
    crop = y[0, 0].cpu().numpy()[224:288, 224:288]
    assert np.abs(crop - g["f512_he_crop"]).max() <= 2e-5
