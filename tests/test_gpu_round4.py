"""Round 4: what the default inference mode rests on, and the training paths the published runs use.
  * the MAE gate on a SECOND weight set: unet_2 trained for a few hundred steps by this package's own trainer (the reference ships no UNet
    checkpoint, .MISSING_LARGE_BLOBS:7-12, so "trained-like" is the closest pin available) -- policy next to DEFAULT_MODE (model/__init__.py):
    a default must keep >= 3x margin to the north star's 1e-4 on BOTH weight sets;
  * a train step of the published 'dropout' run's configuration (models/unet/dropout/*/config.json: drop_rate 0.1, loss 'l1', covers only) with
    an explicit keep-mask, gradients against the oracle's autograd (unet.py:32-42,139-140), in exact f32 and in the default planar arithmetic."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, gpu_model, images01
from ws_unet_amd import formula, losses
from ws_unet_amd.model import get_model
from oracle import unet_ref

pytestmark = pytest.mark.gpu


def rel_l2(got, ref) -> float:
    return float((got.double() - ref.double()).norm() / ref.double().norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def trained_state():
    """unet_2 from the PyTorch-default-like formula init after 300 AdamW steps (L1WS, lr 1e-3) in the default planar training arithmetic."""
    from ws_unet_amd.trainer import synthetic_pretrain
    m = gpu_model(2, "default", None)
    first = synthetic_pretrain(m, steps=1)
    last = synthetic_pretrain(m, steps=299)
    assert last < 0.5 * first, (first, last)                            # it did learn to predict pixels
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


def test_mae_gate_on_trained_weights(trained_state):
    """f16f4p (default) <= 3e-5 and f16f8p <= 1e-5 MAE of the [0,1] output against the fp32 CPU oracle on trained-like weights, on images the
    training never saw (the 'he' formula weights carry the other half of the gate: test_gpu_forward.py::test_mae_gate_512_batch)."""
    _, x = images01(4, 256, 256, seed=77)
    sd = {k: v.float() for k, v in trained_state.items()}
    with torch.no_grad():
        ref = unet_ref.unet_forward(x.clone(), sd, 2)
    assert float(ref.std()) > 0.05, float(ref.std())                    # the output uses its range (default init alone gives 0.504 +- 2e-4)
    got = {}
    for mode, band in (("f16f4p", 3e-5), ("f16f8p", 1e-5), ("f32", 1e-6)):          # measured: 3.1e-6, 4.7e-7, 2.9e-8 (the bands are VERDICT r03 next #2a)
        m = get_model("unet_2", in_channels=1, out_channels=1, channel=[0], drop_rate=None, mode=mode)
        m.load_state_dict(trained_state)
        m = m.to(DEV)
        with torch.no_grad():
            y = m(x.to(DEV)).cpu()
        assert m.mode == mode                                           # no range fallback happened
        got[mode] = float((y - ref).abs().mean())
        assert got[mode] <= band, (mode, got)
    print("MAE on trained-like weights:", got)


@pytest.mark.parametrize("mode", ["f32", None])
def test_train_step_of_the_dropout_config(mode):
    """The published 'dropout' run: UniformDropout(p = 0.1) on the input, loss 'l1', covers only (inputs == covers, alphas 0).  One forward +
    backward with an explicit keep-mask against the oracle's autograd: exact f32, and the default mode's planar training arithmetic."""
    ns, n, size = 2, 2, 64
    cov_u8 = formula.synthetic_images(n, size, size, seed=21)
    covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None]
    mask = (torch.rand((n, 1, size, size), generator=torch.Generator().manual_seed(9)) < 0.9).float()
    assert 0.85 < float(mask.mean()) < 0.95
    # oracle: the reference's layers under torch autograd on the CPU
    sd_np = formula.formula_state_dict(ns, "he")
    ref = unet_ref.build_ref(ns, sd_np)
    out_ref = ref(covers.clone(), dropout_mask=mask)
    loss_ref = (out_ref - covers).abs().mean()
    loss_ref.backward()
    gref = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    # product
    model = gpu_model(ns, "he", mode, drop_rate=0.1)
    if mode is None:
        assert model.train_mode == "f16f8p" and model.train_products == "f16"
    model.input_dropout.next_mask = mask.to(DEV)
    x = covers.clone().to(DEV)
    out = model(x)
    assert torch.equal(model.input_dropout.mask.cpu(), mask)
    loss = losses.L1Loss()(out, (covers.to(DEV), None), x)
    loss.backward()
    tol_out, tol_loss, band = (2e-6, 1e-6, 5e-4) if mode == "f32" else (1e-4, 1e-4, 5e-3)      # measured worst relative L2: 1.0e-4 (f32), 9.5e-4 (planar: ReLU-mask and sign flips on rounding noise)
    assert float((out.detach().cpu() - out_ref.detach()).abs().max()) <= tol_out
    assert abs(loss.item() - loss_ref.item()) <= tol_loss * abs(loss_ref.item()) + 1e-9
    worst = {}
    for k, p in model.named_parameters():
        worst[k] = rel_l2(p.grad.detach().cpu(), gref[k])
        assert worst[k] < band, (k, worst[k])
    print("dropout-config gradients, mode", mode, "worst rel L2", max(worst.values()))


@pytest.mark.parametrize("cin,cout", [(3, 1), (2, 3)])
def test_multi_plane_inputs_in_the_default_mode(cin, cout):
    """VERDICT r03 missing #6: a whole unet_2 with in_channels > 1 (src/_defs/loader.py:61-103 stacks colour planes; the published runs use one)
    and several output planes in the DEFAULT mode against the CPU oracle: the first-layer kernel reads up to 8 planes, the fused head writes up to 4."""
    sd_np = formula.formula_state_dict(2, "he", in_channels=cin, out_channels=cout)
    m = get_model("unet_2", in_channels=cin, out_channels=cout, channel=[0], drop_rate=None, mode=None)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    m = m.to(DEV)
    x = torch.rand((2, cin, 64, 96), generator=torch.Generator().manual_seed(31))
    with torch.no_grad():
        y = m(x.to(DEV)).cpu()
        ref = unet_ref.unet_forward(x.clone(), unet_ref.to_torch_state(sd_np), 2)
    assert m.mode == "f16f4p" and tuple(y.shape) == (2, cout, 64, 96)
    assert float((y - ref).abs().max()) <= 6e-4 and float((y - ref).abs().mean()) <= 5e-5, (float((y - ref).abs().max()), float((y - ref).abs().mean()))
