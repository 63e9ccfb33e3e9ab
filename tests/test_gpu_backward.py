"""GPU parity tests of the backward pass, loss and optimiser: every K7/K8/K9 kernel against torch autograd on
the CPU oracle, whole-network gradients and AdamW trajectories against golden vectors from the reference."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import DEV, to_nhwc, from_nhwc, rand_act, images01, gpu_model
from ws_unet_amd import formula, ops, losses
from ws_unet_amd.trainer import FlatAdamW, Trainer, create_run_name
from oracle import unet_ref, losses_ref

pytestmark = pytest.mark.gpu

F32 = ops.mode_id("f32")


def close(got, ref, rtol, what):
    got, ref = got.double().cpu(), ref.double().cpu()
    err = (got - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-30)
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.2e} > {rtol:.0e})"


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
@pytest.mark.parametrize("shape", [
    (2, 8, 32, 64, 0, 64),        # exact tile
    (1, 12, 40, 64, 64, 128),     # fused concat (two gradient outputs), ragged
    (1, 3, 5, 64, 0, 64),         # H = 3: rows -1 and H both fold onto row 1
    (1, 2, 2, 128, 0, 64),        # smallest legal image
    (2, 20, 36, 128, 0, 128),
])
def test_conv3x3_backward_vs_autograd(mode, shape):
    n, h, w, c1, c2, cout = shape
    m = ops.mode_id(mode)
    x1 = rand_act((n, c1, h, w), f"b/x1/{shape}").requires_grad_(True)
    x2 = rand_act((n, c2, h, w), f"b/x2/{shape}").requires_grad_(True) if c2 else None
    wt = torch.from_numpy(formula.formula_tensor(f"b/w/{shape}", (cout, c1 + c2, 3, 3), (6.0 / (9 * (c1 + c2))) ** 0.5)).requires_grad_(True)
    b = torch.from_numpy(formula.formula_tensor(f"b/b/{shape}", (cout,), 0.1)).requires_grad_(True)
    g = torch.from_numpy(formula.formula_tensor(f"b/g/{shape}", (n, cout, h, w), 1.0))
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    y = unet_ref.conv3x3_reflect(xin, wt, b)
    y.backward(g)
    gd = to_nhwc(g, "f32")
    dw, db = ops.conv3x3_bwd_weight(gd, to_nhwc(x1.detach(), "f32"), None if x2 is None else to_nhwc(x2.detach(), "f32"), mode=m)
    close(dw, wt.grad, 2e-5 if mode == "f32" else 5e-5, "dW"); close(db, b.grad, 2e-5, "db")
    wd = wt.detach().to(DEV)
    # masks: x1 itself (post-ReLU activations -> zeros where x1 == 0), none for x2's... both variants are exercised
    mask1 = to_nhwc(x1.detach(), "f32")
    dx1, dx2 = ops.conv3x3_bwd_data(gd, ops.pack_conv3x3(wd, m, dgrad=True), wd, c1, mask1, None, m)
    tol = 2e-5 if mode == "f32" else 1e-4
    close(from_nhwc(dx1), x1.grad * (x1.detach() > 0), tol, "dx1 (masked)")
    if x2 is not None:
        close(from_nhwc(dx2), x2.grad, tol, "dx2 (unmasked)")
    dx1u, _ = ops.conv3x3_bwd_data(gd, ops.pack_conv3x3(wd, m, dgrad=True), wd, c1, None, None, m)
    close(from_nhwc(dx1u), x1.grad, tol, "dx1 (unmasked)")


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
@pytest.mark.parametrize("shape", [(2, 4, 32, 128, 64), (1, 5, 37, 256, 128), (1, 1, 1, 64, 64)])
def test_convt2x2_backward_vs_autograd(mode, shape):
    n, h, w, cin, cout = shape
    m = ops.mode_id(mode)
    x = rand_act((n, cin, h, w), f"ctb/x/{shape}").requires_grad_(True)
    wt = torch.from_numpy(formula.formula_tensor(f"ctb/w/{shape}", (cin, cout, 2, 2), (6.0 / cin) ** 0.5)).requires_grad_(True)
    b = torch.from_numpy(formula.formula_tensor(f"ctb/b/{shape}", (cout,), 0.1)).requires_grad_(True)
    dy = torch.from_numpy(formula.formula_tensor(f"ctb/dy/{shape}", (n, cout, 2 * h, 2 * w), 1.0))
    F.conv_transpose2d(x, wt, b, stride=2).backward(dy)
    dyd, xd = to_nhwc(dy, "f32"), to_nhwc(x.detach(), "f32")
    dw, db = ops.convt2x2_bwd_weight(xd, dyd, mode=m)
    close(dw, wt.grad, 2e-5 if mode == "f32" else 5e-5, "convT dW"); close(db, b.grad, 2e-5, "convT db")
    dx = ops.convt2x2_bwd_data(dyd, ops.pack_convt2x2_dgrad(wt.detach().to(DEV), m), cin, xd, m)
    close(from_nhwc(dx), x.grad * (x.detach() > 0), 2e-5 if mode == "f32" else 1e-4, "convT dx (masked)")


def test_pool_head_first_backward_vs_autograd():
    # max-pool backward with ties (post-ReLU zeros) + skip accumulation + ReLU mask
    z = rand_act((2, 64, 12, 20), "pb/z", relu=False).requires_grad_(True)
    a = F.relu(z)
    p = F.max_pool2d(a, 2, 2)
    dyp = torch.from_numpy(formula.formula_tensor("pb/dy", tuple(p.shape), 1.0))
    skip = torch.from_numpy(formula.formula_tensor("pb/skip", tuple(a.shape), 1.0))
    (p * dyp).sum().backward()
    ad = to_nhwc(a.detach(), "f32")
    yp, idx = ops.maxpool2x2(ad, F32, want_idx=True)
    g0 = ops.maxpool2x2_bwd(None, to_nhwc(dyp, "f32"), idx, yp)
    np.testing.assert_array_equal(from_nhwc(g0).numpy(), z.grad.numpy())          # exact routing, relu'(0) = 0
    skip_masked = to_nhwc(skip * (a.detach() > 0), "f32")
    g1 = ops.maxpool2x2_bwd(skip_masked.clone(), to_nhwc(dyp, "f32"), idx, yp)
    np.testing.assert_allclose(from_nhwc(g1).numpy(), (z.grad + skip * (a.detach() > 0)).numpy(), rtol=0, atol=1e-6)
    # head
    for cout in (1, 3):
        x = rand_act((2, 64, 9, 11), f"hb/x{cout}").requires_grad_(True)
        wt = torch.from_numpy(formula.formula_tensor(f"hb/w{cout}", (cout, 64, 1, 1), 0.4)).requires_grad_(True)
        b = torch.from_numpy(formula.formula_tensor(f"hb/b{cout}", (cout,), 0.1)).requires_grad_(True)
        dout = torch.from_numpy(formula.formula_tensor(f"hb/d{cout}", (2, cout, 9, 11), 1.0))
        out = torch.sigmoid(F.conv2d(x, wt, b))
        out.backward(dout)
        gx, dw, db = ops.conv1x1_sigmoid_bwd(to_nhwc(x.detach(), "f32"), wt.detach().to(DEV), out.detach().to(DEV), dout.to(DEV))
        close(from_nhwc(gx), x.grad * (x.detach() > 0), 2e-5, "head gx"); close(dw, wt.grad, 2e-5, "head dw"); close(db, b.grad, 2e-5, "head db")
    # first layer weight gradient
    for cin in (1, 3):
        x = rand_act((2, cin, 14, 18), f"fb/x{cin}", relu=False)
        wt = torch.from_numpy(formula.formula_tensor(f"fb/w{cin}", (64, cin, 3, 3), 0.5)).requires_grad_(True)
        b = torch.zeros(64, requires_grad=True)
        g = torch.from_numpy(formula.formula_tensor(f"fb/g{cin}", (2, 64, 14, 18), 1.0))
        unet_ref.conv3x3_reflect(x, wt, b).backward(g)
        dw, db = ops.conv3x3_first_bwd_weight(to_nhwc(g, "f32"), x.to(DEV))
        close(dw, wt.grad, 2e-5, "first dW"); close(db, b.grad, 2e-5, "first db")


def _loss_inputs():
    cov_u8 = formula.synthetic_images(4, 32, 32, seed=31)
    st_u8 = cov_u8.copy()
    st_u8[0] = formula.lsbr_embed(cov_u8[0], 0.4, seed=1)
    st_u8[2] = formula.lsbr_embed(cov_u8[2], 1.0, seed=2)
    covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None]
    inputs = torch.from_numpy(st_u8.astype(np.float32) / np.float32(255.))[:, None]
    return covers, inputs, torch.tensor([0.4, 0.0, 1.0, 0.0])


def test_losses_golden(golden):
    g = golden["losses"]
    covers, inputs, alphas = _loss_inputs()
    for i, (name, cls) in enumerate([("l1", losses.L1Loss), ("l2", losses.L2Loss), ("ws", losses.WSLoss), ("l1ws", losses.L1WSLoss)]):
        outputs = torch.from_numpy(g["loss_outputs"]).to(DEV).requires_grad_(True)
        crit = cls()
        v = crit(outputs, (covers.to(DEV), alphas.to(DEV)), inputs.to(DEV))
        v.backward()
        ref = float(g["loss_values"][{"l1": 0, "l2": 1, "ws": 2, "l1ws": 3}[name]])
        assert math.isclose(v.item(), ref, rel_tol=2e-6), name
        np.testing.assert_allclose(outputs.grad.cpu().numpy(), g[f"loss_{name}_dout"], rtol=1e-5, atol=1e-10, err_msg=name)
    with pytest.raises(NotImplementedError):
        losses.get_loss("crossentropy")


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
@pytest.mark.parametrize("ns", [0, 1, 2])
def test_unet_gradients_golden(golden, ns, mode):
    """dL/dtheta of L1WS on 2x1x64x64 cover/stego pairs for all parameter tensors (reference autograd golden); an 'f32' model
    trains in exact fp32, the default 'bf16x3' model with split-bf16 arithmetic."""
    g = golden["unet_grad"]
    model = gpu_model(ns, "he", mode)
    assert model.train_mode == mode
    cov_u8 = formula.synthetic_images(2, 64, 64, seed=11)
    st_u8 = cov_u8.copy(); st_u8[0] = formula.lsbr_embed(cov_u8[0], 0.4, seed=5)
    covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None].to(DEV)
    inputs = torch.from_numpy(st_u8.astype(np.float32) / np.float32(255.))[:, None].to(DEV).requires_grad_(True)
    alphas = torch.tensor([0.4, 0.0], device=DEV)
    out = model(inputs)
    assert out.requires_grad
    loss = losses.L1WSLoss()(out, (covers, alphas), inputs)
    loss.backward()
    assert math.isclose(loss.item(), float(g[f"grad{ns}_loss"][0]), rel_tol=1e-5)
    # the matrix layers of a 'bf16x3' training forward run the f16f8 arithmetic on fp32 tensors (model.train_fwd_mode = 'f16f8x')
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"grad{ns}_out"], atol=4e-6 if mode == "f32" else 1e-4, rtol=0)
    for k, p in model.named_parameters():
        got = p.grad.detach().cpu().numpy().reshape(-1)
        ref = g[f"grad{ns}_{k}_sub"]
        if not (ns == 0 or got.size <= 4096):
            got = got[::97]
        # tolerance: two fp32 implementations of this loss differ by ~1e-4..1e-3 of the gradient scale, because
        # sign(cover - out) and the ReLU masks flip on fp32 rounding noise.  Measured against an fp64 oracle
        # (tools/diag_grads.py, profiles/r01/grad_error_vs_fp64_unet2_64x64.txt): torch-CPU fp32 is off by 1e-4
        # relative L2 on every layer, libwsu by 2e-5..3e-4.
        # The split-bf16 forward moves the outputs by ~2e-6, which flips more of those signs: against fp64 its gradients are off by
        # 1-2.5e-4 relative L2 on unet_2 (2x torch-CPU fp32) and by 3e-3 / 1e-2 of the scale at worst on the ill-conditioned
        # unet_0 case, where fp32 itself misses the fp64 loss by 2e-4 (profiles/r01/grad_error_vs_fp64_bf16x3.txt).
        scale = float(np.abs(ref).max())
        tol = 1.5e-3 if mode == "f32" else 1.5e-2
        np.testing.assert_allclose(got, ref, rtol=0, atol=tol * scale + 1e-12, err_msg=f"unet_{ns} {k}")
        s = g[f"grad{ns}_{k}_sum"]
        full = p.grad.detach().double().cpu().numpy()
        assert math.isclose(float(np.sqrt((full ** 2).sum())), s[2], rel_tol=1e-3 if mode == "f32" else 8e-3), k
    # inputs.requires_grad exercises the input-gradient kernel here, but the golden grad{ns}_dx is not comparable: the
    # reference's WS term depends on `inputs` directly (losses.py:59-62) and autograd adds that path, while the fused loss
    # treats inputs as data (as the training loop does).  The input gradient is checked in test_saliency_style_input_gradient.
    assert inputs.grad is not None and torch.isfinite(inputs.grad).all()
    # gradients are deterministic (no float atomics): bitwise equal on a repeat
    first = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad()
    inputs.grad = None
    losses.L1WSLoss()(model(inputs), (covers, alphas), inputs).backward()
    for k, p in model.named_parameters():
        assert torch.equal(p.grad, first[k]), k


def test_adamw_and_train_step_golden(golden):
    """3 AdamW steps on unet_0 with the L1WS loss reproduce torch.optim.AdamW on the reference model."""
    g = golden["adamw"]
    model = gpu_model(0, "he", "f32")
    cov_u8 = formula.synthetic_images(2, 32, 32, seed=41)
    st_u8 = cov_u8.copy(); st_u8[1] = formula.lsbr_embed(cov_u8[1], 0.4, seed=9)
    covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None].to(DEV)
    inputs = torch.from_numpy(st_u8.astype(np.float32) / np.float32(255.))[:, None].to(DEV)
    alphas = torch.tensor([0.0, 0.4], device=DEV)
    tr = Trainer(model, loss="l1ws", lr=1e-4)
    ls = [tr.train_step(inputs.clone(), covers, alphas)[0].item() for _ in range(3)]
    np.testing.assert_allclose(ls, g["adamw_losses"], rtol=2e-5)
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"adamw_{k}"], rtol=0, atol=5e-7, err_msg=k)
    # the same model driven by torch.optim.AdamW through plain autograd gives the same trajectory
    model2 = gpu_model(0, "he", "f32")
    opt = torch.optim.AdamW(model2.parameters(), 1e-4)
    crit = losses.L1WSLoss()
    for _ in range(3):
        opt.zero_grad()
        crit(model2(inputs.clone()), (covers, alphas), inputs).backward()
        opt.step()
    for (k, p), (_, q) in zip(model.named_parameters(), model2.named_parameters()):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=0, atol=5e-7, err_msg=k)


@pytest.mark.parametrize("mode", ["f32", None])
def test_training_reduces_loss_and_checkpoints(tmp_path, mode):
    model = gpu_model(1, "default", mode)                       # None: the default inference mode, trained in its planar arithmetic
    cfg = {"network": "unet_1", "alpha": "0.400", "grayscale": True, "loss": "l1ws", "loss_lambda": 0.25,
           "learning_rate": 0.0001, "drop_rate": 0.0, "demosaic": None, "demosaic_oracle": False, "channel": [0]}
    assert create_run_name({**cfg, "network": "unet_2"}) == "unet_2-alpha_0.400_grayscale_l1ws_0.25_lr_0.0001_"   # published run dir suffix
    tr = Trainer(model, loss="l1ws", lr=1e-3, out_dir=tmp_path / "run", config=cfg, patience=2)
    cov_u8 = formula.synthetic_images(8, 64, 64, seed=61)
    st_u8 = np.stack([formula.lsbr_embed(c, 0.4, seed=i) if i % 2 else c for i, c in enumerate(cov_u8)])
    covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None]
    inputs = torch.from_numpy(st_u8.astype(np.float32) / np.float32(255.))[:, None]
    alphas = torch.tensor([0.4 if i % 2 else 0.0 for i in range(8)])
    loader = [(inputs[i:i + 4], (covers[i:i + 4], alphas[i:i + 4])) for i in (0, 4)]
    tr.fit(loader, loader, num_epochs=4)
    tags = {t for _, t, _ in tr.scalars}
    assert tags == {"train/loss", "train/mae", "train/ws", "train/skipped_steps", "val/loss", "val/mae", "val/ws"}
    assert tr.skipped_steps() == 0
    tl = [v for e, t, v in tr.scalars if t == "train/loss"]
    assert tl[-1] < tl[0]
    ck = torch.load(tmp_path / "run" / "model" / "best_model.pt.tar", weights_only=True)
    assert set(ck) == {"epoch", "state_dict", "best_val_loss", "patience", "optimizer", "scheduler"}
    assert list(ck["state_dict"]) == list(model.state_dict())
    assert (tmp_path / "run" / "config.json").exists() and (tmp_path / "run" / "log" / "scalars.csv").exists()
    # the device-side epoch meters (wsu_ws_meter_beta + the fused loss kernel's L1 sum) equal the reference's numpy meters
    from ws_unet_amd import metrics
    mae, wsm = metrics.MAEMeter(multiplier=1), metrics.WSMeter()
    with torch.no_grad():
        for x, (c, a) in loader:
            o = model(x.to(DEV)).cpu().numpy()
            mae.update(c.numpy(), o)
            wsm.update(x.numpy(), o, a.numpy())
    val = tr._run_epoch(loader, False, 99)
    got = {t: v for e, t, v in tr.scalars if e == 99}
    assert abs(got["val/mae"] - mae.avg) <= 1e-6 * max(1.0, abs(mae.avg))
    assert abs(got["val/ws"] - wsm.avg) <= 1e-6 * max(1.0, abs(wsm.avg)) + 1e-9
    assert val == got["val/loss"]


def test_saliency_style_input_gradient():
    """src/saliency.py:133-174: parameters frozen, input requires grad, backward from ONE output pixel; the gradient
    is confined to the receptive field and matches the CPU oracle."""
    model = gpu_model(2, "he", "f32")
    model.input_dropout = None
    for p in model.parameters():
        p.requires_grad = False
    _, x = images01(1, 64, 64, seed=13)
    xd = x.to(DEV).requires_grad_(True)
    out = model(xd)
    out[0, 0, 30, 33].backward()
    gx = xd.grad.cpu()
    ref_m = unet_ref.build_ref(2, formula.formula_state_dict(2, "he"))
    xr = x.clone().requires_grad_(True)
    ref_m(xr)[0, 0, 30, 33].backward()
    scale = xr.grad.abs().max().item()
    assert scale > 0
    np.testing.assert_allclose(gx.numpy(), xr.grad.numpy(), rtol=0, atol=2e-4 * scale)
    nz = (xr.grad[0, 0] != 0).nonzero()
    assert nz[:, 0].min() >= 30 - 24 and nz[:, 0].max() <= 30 + 24           # receptive field of unet_2 ~ +-22
    with torch.no_grad():
        assert model(torch.zeros(0, 1, 64, 64, device=DEV)).shape == (0, 1, 64, 64)     # empty batch


def test_pair_loader_feeds_the_trainer_from_png_files(tmp_path):
    """PNG pairs on disk -> libwsu_io decode -> uint8 upload -> wsu_u8_to_unit_f32 -> Trainer.fit; the device batches equal the
    host-logic (uint8) batches / 255 exactly."""
    from PIL import Image
    from ws_unet_amd.data.pairs import PairLoader
    (tmp_path / "images").mkdir()
    sd = tmp_path / "stego_LSBR_alpha_0.4"
    sd.mkdir()
    u8 = formula.synthetic_images(6, 64, 64, seed=88)
    for i in range(6):
        Image.fromarray(u8[i]).save(tmp_path / "images" / f"{i}.png")
        Image.fromarray(formula.lsbr_embed(u8[i], 0.4, seed=i)).save(sd / f"{i}.png")
    (tmp_path / "images" / "files.csv").write_text("name,height,width\n" + "".join(f"images/{i}.png,64,64\n" for i in range(6)))
    (sd / "files.csv").write_text("name,height,width,stego_method,alpha\n" + "".join(f"stego_LSBR_alpha_0.4/{i}.png,64,64,LSBR,0.4\n" for i in range(6)))
    host = list(PairLoader(tmp_path, None, "LSBR", 0.4, batch_size=4, seed=3))
    dev = PairLoader(tmp_path, None, "LSBR", 0.4, batch_size=4, seed=3, device=torch.device(DEV))
    got = list(dev)
    assert len(got) == len(host) == 3
    for (x, (c, a)), (xh, (ch, ah)) in zip(got, host):
        assert x.is_cuda and x.shape == (4, 1, 64, 64) and x.dtype == torch.float32
        np.testing.assert_array_equal(x[:, 0].cpu().numpy(), xh.numpy().astype(np.float32) / np.float32(255.))
        np.testing.assert_array_equal(c[:, 0].cpu().numpy(), ch.numpy().astype(np.float32) / np.float32(255.))
        assert torch.equal(a.cpu(), ah)
    model = gpu_model(1, "default", "f32")
    tr = Trainer(model, loss="l1ws", lr=1e-3, patience=5)
    tr.fit(dev, dev, num_epochs=3)
    tl = [v for e, t, v in tr.scalars if t == "train/loss"]
    assert len(tl) == 3 and tl[-1] < tl[0] and all(np.isfinite(tl))


@pytest.mark.parametrize("mode", ["f32", None])
def test_train_driver_replays_a_published_style_config(tmp_path, mode):
    """ws_unet_amd.train.train(): run directory layout, config.json keys, checkpoints, weight-only resume (detector/train.py:143-304) -- in
    exact f32 and with `mode` unset: the default inference mode and its planar training arithmetic (what bench.py's train_step leg times)."""
    import json
    from PIL import Image
    from ws_unet_amd import train as train_mod
    data = tmp_path / "data"
    (data / "images").mkdir(parents=True)
    sd = data / "stego_LSBR_alpha_0.4_independent_images"
    sd.mkdir()
    u8 = formula.synthetic_images(6, 64, 64, seed=91)
    rows = []
    for i in range(6):
        Image.fromarray(u8[i]).save(data / "images" / f"{i}.png")
        Image.fromarray(formula.lsbr_embed(u8[i], 0.4, seed=i)).save(sd / f"{i}.png")
    hdr = "name,height,width,channels,device,stego_method,simulator,alpha,demosaic,color,color_strategy,beta_hat\n"
    def split(ids):
        return hdr + "".join(f"images/{i}.png,64,64,,,,,,,,,\n" for i in ids) + "".join(
            f"stego_LSBR_alpha_0.4_independent_images/{i}.png,64,64,,,LSBR,mi,0.4,,,independent,\n" for i in ids)
    (data / "split_tr.csv").write_text(split([0, 1, 2, 3]))
    (data / "split_va.csv").write_text(split([4, 5]))
    cfg = {"dataset": str(data), "output_dir": str(tmp_path / "runs"), "network": "unet_1", "stego_method": "LSBR", "alpha": "0.400",
           "loss": "l1ws", "batch_size": 4, "num_epochs": 2, "patience": 5, "learning_rate": 1e-3, "drop_rate": 0.0, "seed": 7,
           "SLURM_JOB_ID": "42", **({"mode": mode} if mode else {})}
    best = train_mod.train(cfg)
    runs = list((tmp_path / "runs" / "LSBR").iterdir())
    assert len(runs) == 1 and runs[0].name.split("-", 2)[1] == "42"
    assert runs[0].name.endswith("unet_1-alpha_0.400_grayscale_l1ws_0.25_lr_0.001_")
    saved = json.loads((runs[0] / "config.json").read_text())
    assert saved["network"] == "unet_1" and saved["tr_csv"] == "split_tr.csv" and saved["batch_size"] == 4 and "mode" not in saved
    for f in ("model/latest_model.pt.tar", "model/best_model.pt.tar", "log/scalars.csv"):
        assert (runs[0] / f).exists(), f
    assert np.isfinite(best)
    # the evaluate-side discovery finds the run, and a second run can resume from its weights
    from ws_unet_amd import evaluate
    assert evaluate.get_model_name("LSBR", model_dir=tmp_path / "runs") == runs[0].name
    best2 = train_mod.train({**cfg, "resume": runs[0].name, "num_epochs": 1, "experiment_dir_suffix": "ft"})
    assert np.isfinite(best2) and len(list((tmp_path / "runs" / "LSBR").iterdir())) == 2
    with pytest.raises(Exception, match="no checkpoint"):
        train_mod.train({**cfg, "resume": "missing-run"})


def test_train_step_helpers_and_finite_guard():
    """wsu_pow2_grad_scale / wsu_scale_* against their torch definitions, and the finite guard: a poisoned gradient bucket leaves parameters
    and AdamW moments untouched, the skipped step is counted, the next clean step proceeds (an inf must not reach the moments)."""
    g = torch.Generator(device=DEV).manual_seed(3)
    for mag in (3e-7, 1.0, 517.0, 0.0):
        x = (torch.randn(3, 1, 33, 65, device=DEV, generator=g) * mag).contiguous()
        s2 = ops.pow2_grad_scale(x)
        ref = torch.exp2(torch.floor(2.0 - torch.log2(x.abs().max().clamp_min(1e-30))))
        assert s2[0].item() == ref.item() and s2[0].item() * s2[1].item() == 1.0
        if mag:
            assert 2.0 < (x.abs().max() * s2[0]).item() <= 4.0
        assert torch.equal(ops.scale_by(x, s2[0:1]), x * s2[0])
    ts = [torch.randn(n, device=DEV, generator=g) for n in (5, 1024, 1025, 70000)]
    want = [t * 0.125 for t in ts]
    ops.scale_many_(ts + [None], torch.tensor([0.125], device=DEV))
    assert all(torch.equal(a, b) for a, b in zip(ts, want))

    model = gpu_model(0, "he", "f32")
    cov_u8 = formula.synthetic_images(2, 32, 32, seed=41)
    covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None].to(DEV)
    alphas = torch.tensor([0.0, 0.0], device=DEV)
    tr = Trainer(model, loss="l1ws", lr=1e-3)
    tr.train_step(covers.clone(), covers, alphas)
    before = tr.opt.flat_param.clone(); m0, v0 = tr.opt.exp_avg.clone(), tr.opt.exp_avg_sq.clone()
    bad = covers.clone(); bad[0, 0, 3, 3] = float("inf")
    tr.train_step(bad, covers, alphas)                                   # inf input -> NaN / inf gradients
    assert not torch.isfinite(tr.opt.flat_grad).all()
    assert torch.equal(tr.opt.flat_param, before) and torch.equal(tr.opt.exp_avg, m0) and torch.equal(tr.opt.exp_avg_sq, v0)
    assert tr.skipped_steps() == 1
    tr.train_step(covers.clone(), covers, alphas)
    assert tr.skipped_steps() == 1 and not torch.equal(tr.opt.flat_param, before) and torch.isfinite(tr.opt.flat_param).all()
