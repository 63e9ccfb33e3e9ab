#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference, read-only):

    python tests/golden/make_golden.py

What is imported from the reference:
  * src/unet/model            -> get_model / UNet / UniformDropout   (arithmetic oracle, fwd + autograd bwd)
  * src/_defs/losses.py       -> L1Loss / WSLoss / L1WSLoss          (by file path; a dummy `timm` module is
  * src/_defs/metrics.py      -> MAEMeter / WSMeter                   registered because losses.py:2 imports it unused)
  * src/fabrika.py            -> precovers / stego_spatial iterate
  * src/ws/estimate.py        -> attack, NAMED_FILTERS;  src/filters/evaluate.py -> get_filter_estimator;
    src/_defs/filters.py      -> get_processor_2d                    (placeholders for seaborn/conseal/_defs/unet, see gen_ws_attack)
Everything written is DATA (inputs are formula-generated, see ws_unet_amd/formula.py;
outputs are arrays / JSON).  No reference source text is stored.
"""
import importlib.util
import json
import os
import shutil
import sys
import tempfile
import types
from pathlib import Path
from unittest import mock

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(REF / "src" / "unet"))

from ws_unet_amd import formula  # noqa: E402
from model import get_model      # noqa: E402  (reference)

torch.set_num_threads(8)


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_model(nsteps, variant, drop_rate=None, seed=0):
    m = get_model(f"unet_{nsteps}", in_channels=1, out_channels=1, channel=[0], drop_rate=drop_rate)
    sd = formula.formula_state_dict(nsteps, variant, seed)
    assert list(sd.keys()) == list(m.state_dict().keys()), "state_dict key order differs from reference"
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m


def images01(n, h, w, seed):
    u8 = formula.synthetic_images(n, h, w, seed)
    return u8, torch.from_numpy(u8.astype(np.float32) / np.float32(255.))[:, None]


def capture_intermediates(m, x):
    """Post-activation outputs of every layer via forward hooks (conv outputs get ReLU applied,
    matching the names in unet.py:141-186)."""
    outs = {}
    hooks = []
    for name, mod in m.named_modules():
        if name and name != "input_dropout":
            hooks.append(mod.register_forward_hook(lambda _m, _i, o, name=name: outs.__setitem__(name, o.detach().clone())))
    with torch.no_grad():
        y = m(x)
    for h in hooks:
        h.remove()
    named = {}
    for k, v in outs.items():
        if k.startswith("pool"):
            named["xp" + k[-1]] = v
        elif k.startswith("upconv"):
            named["xu" + k[-1]] = v
        elif k == "outconv":
            named["logit"] = v
        else:
            named["x" + k] = torch.relu(v)
    return y, named


def gen_forward_small(out):
    for ns in range(5):
        for variant in (("he", "default") if ns == 2 else ("he",)):
            m = ref_model(ns, variant)
            m.input_dropout = None
            _, x = images01(2, 32, 32, seed=1)
            with torch.no_grad():
                y = m(x.clone())
            out[f"y_unet{ns}_{variant}"] = y.numpy()
    # intermediates: unet_2, he, N=1 32x32; channels subsampled [::8] + float64 sums
    m = ref_model(2, "he"); m.input_dropout = None
    _, x = images01(1, 32, 32, seed=2)
    y, named = capture_intermediates(m, x.clone())
    out["inter_y"] = y.numpy()
    for k, v in named.items():
        a = v.numpy()
        out[f"inter_{k}_sub"] = a[:, ::8].copy()
        out[f"inter_{k}_sum"] = np.array([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()])
    # non-square / non-multiple-of-tile shape: unet_2 on 2x1x24x40
    m = ref_model(2, "he"); m.input_dropout = None
    _, x = images01(2, 24, 40, seed=3)
    with torch.no_grad():
        out["y_unet2_he_24x40"] = m(x.clone()).numpy()


def gen_forward_512(out):
    for variant in ("he", "default"):
        m = ref_model(2, variant); m.input_dropout = None
        _, x = images01(1, 512, 512, seed=7)
        with torch.no_grad():
            y = m(x.clone()).numpy()[0, 0]
        out[f"f512_{variant}_tilesum"] = y.astype(np.float64).reshape(8, 64, 8, 64).sum(axis=(1, 3))
        out[f"f512_{variant}_crop"] = y[224:288, 224:288].copy()
        out[f"f512_{variant}_border"] = np.stack([y[0], y[511], y[:, 0], y[:, 511]])
        out[f"f512_{variant}_stats"] = np.array([y.astype(np.float64).mean(), np.abs(y).max(), y.astype(np.float64).std()])


def gen_grads(out, losses):
    crit = losses.L1WSLoss()
    for ns in (0, 1, 2):
        m = ref_model(ns, "he"); m.input_dropout = None
        cov_u8 = formula.synthetic_images(2, 64, 64, seed=11)
        stego_u8 = cov_u8.copy()
        stego_u8[0] = formula.lsbr_embed(cov_u8[0], 0.4, seed=5)
        covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None]
        inputs = torch.from_numpy(stego_u8.astype(np.float32) / np.float32(255.))[:, None].requires_grad_(True)
        alphas = torch.tensor([0.4, 0.0])
        outputs = m(inputs)
        loss = crit(outputs, (covers, alphas), inputs)
        loss.backward()
        out[f"grad{ns}_loss"] = np.array([loss.item()])
        out[f"grad{ns}_out"] = outputs.detach().numpy()
        out[f"grad{ns}_dx"] = inputs.grad.numpy().copy()
        # the network path alone: the reference's WS term also reaches `inputs` directly (losses.py:59-62), which a fused loss that
        # takes the inputs as data does not differentiate -- feed the loss a detached copy to isolate d loss / d x THROUGH the UNet
        m.zero_grad()
        inputs_net = inputs.detach().clone().requires_grad_(True)
        crit(m(inputs_net), (covers, alphas), inputs_net.detach()).backward()
        out[f"grad{ns}_dx_net"] = inputs_net.grad.numpy().copy()
        m.zero_grad()
        crit(m(inputs), (covers, alphas), inputs).backward()          # restore the parameter gradients of the full loss for the rows below
        for k, p in m.named_parameters():
            g = p.grad.numpy().reshape(-1)
            out[f"grad{ns}_{k}_sum"] = np.array([g.astype(np.float64).sum(), np.abs(g.astype(np.float64)).sum(),
                                                 np.sqrt((g.astype(np.float64) ** 2).sum())])
            out[f"grad{ns}_{k}_sub"] = g.copy() if (ns == 0 or g.size <= 4096) else g[::97].copy()
        # also the plain-L1 run configuration (dropout run: loss 'l1')
        if ns == 1:
            m.zero_grad()
            outputs = m(inputs.detach())
            l1 = losses.L1Loss()(outputs, (covers, alphas))
            l1.backward()
            out["grad1_l1_loss"] = np.array([l1.item()])
            out["grad1_l1_e11.weight"] = m.e11.weight.grad.numpy().copy()
            out["grad1_l1_outconv.weight"] = m.outconv.weight.grad.numpy().copy()


def gen_dropout(out):
    _, x = images01(2, 32, 32, seed=21)
    mask = formula.bernoulli_mask((2, 1, 32, 32), keep_prob=0.9, seed=4242)

    def fake_bernoulli_(self, p=0.5, generator=None):
        assert tuple(self.shape) == mask.shape and abs(float(p) - 0.9) < 1e-12
        self.copy_(torch.from_numpy(mask))
        return self

    m = ref_model(1, "he", drop_rate=0.1)
    with mock.patch.object(torch.Tensor, "bernoulli_", fake_bernoulli_):
        xin = x.clone()
        xd = m.input_dropout(xin.clone())
        with torch.no_grad():
            y = m(xin)            # note: reference mutates xin in place
    out["drop_x_after"] = xd.numpy()
    out["drop_x_inplace"] = xin.numpy()
    out["drop_y_unet1"] = y.numpy()
    # drop_rate=0.: identity, still rewrites input (mask == 1)
    m0 = ref_model(1, "he", drop_rate=0.)
    with torch.no_grad():
        out["drop0_y_unet1"] = m0(x.clone()).numpy()


def gen_micro(out):
    # (a) reflect-border conv on a tiny ramp
    conv = torch.nn.Conv2d(1, 2, kernel_size=3, padding=1, padding_mode="reflect")
    w = formula.formula_tensor("micro/conv.w", (2, 1, 3, 3), 1.0)
    b = formula.formula_tensor("micro/conv.b", (2,), 1.0)
    conv.load_state_dict({"weight": torch.from_numpy(w), "bias": torch.from_numpy(b)})
    x = torch.arange(20, dtype=torch.float32).reshape(1, 1, 4, 5)
    with torch.no_grad():
        out["micro_conv_y"] = conv(x).numpy()
    x2 = torch.arange(4, dtype=torch.float32).reshape(1, 1, 2, 2)     # smallest legal size
    with torch.no_grad():
        out["micro_conv_y_2x2"] = conv(x2).numpy()
    # (b) max-pool ties: forward + backward routing
    xt = torch.tensor([[[[0., 0., 1., 1.], [0., 0., 1., 2.], [3., 3., 0., 5.], [3., 1., 5., 5.]]]], requires_grad=True)
    yt = torch.nn.MaxPool2d(2, 2)(torch.relu(xt))
    (yt * torch.tensor([[[[1., 2.], [3., 4.]]]])).sum().backward()
    out["micro_pool_y"] = yt.detach().numpy()
    out["micro_pool_dx"] = xt.grad.numpy()
    # (c) transposed conv
    ct = torch.nn.ConvTranspose2d(3, 2, kernel_size=2, stride=2)
    wt = formula.formula_tensor("micro/convt.w", (3, 2, 2, 2), 1.0)
    bt = formula.formula_tensor("micro/convt.b", (2,), 1.0)
    ct.load_state_dict({"weight": torch.from_numpy(wt), "bias": torch.from_numpy(bt)})
    xc = torch.from_numpy(formula.formula_tensor("micro/convt.x", (1, 3, 2, 3), 1.0))
    with torch.no_grad():
        out["micro_convt_y"] = ct(xc).numpy()


def gen_losses(out, losses, metrics):
    cov_u8 = formula.synthetic_images(4, 32, 32, seed=31)
    st_u8 = cov_u8.copy()
    st_u8[0] = formula.lsbr_embed(cov_u8[0], 0.4, seed=1)
    st_u8[2] = formula.lsbr_embed(cov_u8[2], 1.0, seed=2)
    covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None]
    inputs = torch.from_numpy(st_u8.astype(np.float32) / np.float32(255.))[:, None]
    alphas = torch.tensor([0.4, 0.0, 1.0, 0.0])
    noise = torch.from_numpy(formula.formula_tensor("loss/noise", (4, 1, 32, 32), 0.02))
    outputs = (covers + noise).clamp(0.001, 0.999).requires_grad_(True)
    vals = {}
    for name, cls in (("l1", losses.L1Loss), ("l2", losses.L2Loss), ("ws", losses.WSLoss), ("l1ws", losses.L1WSLoss)):
        outputs.grad = None
        v = cls()(outputs, (covers, alphas), inputs)
        v.backward()
        vals[name] = v.item()
        out[f"loss_{name}_dout"] = outputs.grad.numpy().copy()
    out["loss_values"] = np.array([vals["l1"], vals["l2"], vals["ws"], vals["l1ws"]])
    out["loss_outputs"] = outputs.detach().numpy()
    mae = metrics.MAEMeter(multiplier=1)
    wsm = metrics.WSMeter()
    for s in (slice(0, 2), slice(2, 4)):
        mae.update(covers.numpy()[s], outputs.detach().numpy()[s])
        wsm.update(inputs.numpy()[s], outputs.detach().numpy()[s], alphas.numpy()[s])
    out["meter_values"] = np.array([mae.avg, wsm.avg])


def gen_adamw(out, losses):
    m = ref_model(0, "he"); m.input_dropout = None
    opt = torch.optim.AdamW(m.parameters(), 1e-4)          # pattern: src/detector/train.py:228
    crit = losses.L1WSLoss()
    cov_u8 = formula.synthetic_images(2, 32, 32, seed=41)
    st_u8 = cov_u8.copy(); st_u8[1] = formula.lsbr_embed(cov_u8[1], 0.4, seed=9)
    covers = torch.from_numpy(cov_u8.astype(np.float32) / np.float32(255.))[:, None]
    inputs = torch.from_numpy(st_u8.astype(np.float32) / np.float32(255.))[:, None]
    alphas = torch.tensor([0.0, 0.4])
    ls = []
    for _ in range(3):
        opt.zero_grad()
        loss = crit(m(inputs.clone()), (covers, alphas), inputs)
        loss.backward()
        opt.step()
        ls.append(loss.item())
    out["adamw_losses"] = np.array(ls)
    for k, p in m.named_parameters():
        out[f"adamw_{k}"] = p.detach().numpy().copy()


def gen_fabrika(fab):
    res = {}

    def echo(fname, **kw):
        return {**kw, "fname": str(fname)}

    cover_it = fab.precovers(iterator="python", convert_to="pandas", ignore_missing=False, n_jobs=-1)(echo)
    stego_it = fab.stego_spatial(iterator="python", convert_to="pandas", ignore_missing=False, n_jobs=-1)(echo)
    data = REF / "data"

    def rows(df):
        return json.loads(df.to_json(orient="records"))

    res["ref_covers"] = rows(cover_it(data))
    res["ref_covers_split_te"] = rows(cover_it(data, split="split_te.csv"))
    res["ref_covers_shuffle3_skip1_take3"] = rows(cover_it(data, shuffle_seed=3, skip_num_images=1, take_num_images=3))
    res["ref_stego_lsbr"] = rows(stego_it(data, stego_method="LSBR"))
    res["ref_stego_lsbr_04"] = rows(stego_it(data, stego_method="LSBR", alpha=0.4))
    res["ref_stego_split_te_hillr"] = rows(stego_it(data, split="split_te.csv", stego_method="HILLR"))
    # synthetic dataset dir (layout definition is repeated in tests/test_fabrika.py)
    tmp = Path(tempfile.mkdtemp())
    try:
        (tmp / "images").mkdir()
        names = [f"images/{i}.png" for i in (1, 10, 11, 2, 20, 3)]
        (tmp / "images" / "files.csv").write_text("name,height,width\n" + "".join(f"{n},512,512\n" for n in names))
        (tmp / "images_b").mkdir()
        (tmp / "images_b" / "files.csv").write_text("name,height,width\nimages_b/7.png,256,256\n")
        sd = tmp / "stego_X_alpha_0.4"
        sd.mkdir()
        sd.joinpath("files.csv").write_text("name,height,width,stego_method,alpha\n" + "".join(
            f"stego_X_alpha_0.4/{i}.png,512,512,X,0.4\n" for i in (1, 10, 2)))
        res["syn_covers"] = rows(cover_it(tmp))
        res["syn_covers_take2"] = rows(cover_it(tmp, take_num_images=2))
        res["syn_covers_shuffle5"] = rows(cover_it(tmp, shuffle_seed=5))
        res["syn_stego"] = rows(stego_it(tmp, stego_method="X", alpha=0.4))
        try:
            stego_it(tmp, stego_method="nope")
            res["syn_stego_empty_error"] = None
        except Exception as e:                                   # fabrika.py:69-70
            res["syn_stego_empty_error"] = str(e)
        for k, v in res.items():
            if isinstance(v, list):
                for r in v:
                    r["fname"] = r["fname"].replace(str(tmp), "<DATASET>").replace(str(data), "<DATASET>")
    finally:
        shutil.rmtree(tmp)
    (HERE / "fabrika.json").write_text(json.dumps(res, indent=1, sort_keys=True))


def gen_png_kat():
    import csv
    kat = {}
    with open(REF / "results" / "prediction" / "filters.csv") as f:
        for r in csv.DictReader(f):
            d = kat.setdefault(r["name"], {})
            if r["mae_3_AVG"]:
                d["mae_3_AVG"] = float(r["mae_3_AVG"])
            if r["mae_3_KB"]:
                d["mae_3_KB"] = float(r["mae_3_KB"])
    (HERE / "filters_kat.json").write_text(json.dumps(kat, indent=1, sort_keys=True))
    shutil.copyfile(REF / "data" / "images" / "10.png", HERE / "cover_10.png")
    os.chmod(HERE / "cover_10.png", 0o644)


def cross_mean(x):
    """Host pixel predictor used for the 'arbitrary callable' WS cases (repeated in tests/test_gpu_ws_attack.py)."""
    return ((x[:-2, 1:-1] + x[2:, 1:-1] + x[1:-1, :-2] + x[1:-1, 2:]) * np.float32(0.25))[..., :1]


def gen_ws_attack(out):
    """Run the reference's own `attack` (src/ws/estimate.py:55-136) and linear predictors (src/filters/evaluate.py).
    Modules those files import at the top but do not use in these functions, and that are absent here (seaborn, conseal)
    or pull in cv2/torchvision (_defs, unet), are registered as empty placeholders for the duration of the import."""
    saved = {k: sys.modules.get(k) for k in ("seaborn", "conseal", "_defs", "filters", "unet")}
    try:
        for k in ("seaborn", "conseal", "filters", "unet"):
            sys.modules[k] = types.ModuleType(k)
        ref_defs_filters = load_by_path("ref_defs_filters", REF / "src" / "_defs" / "filters.py")
        d = types.ModuleType("_defs")
        d.imread4_u8 = d.imread4_f32 = None                      # only default-argument values at def time
        d.get_processor_2d = ref_defs_filters.get_processor_2d
        sys.modules["_defs"] = d
        ref_filters = load_by_path("ref_filters_evaluate", REF / "src" / "filters" / "evaluate.py")
        ref_ws = load_by_path("ref_ws_estimate", REF / "src" / "ws" / "estimate.py")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    process = ref_defs_filters.get_processor_2d(channels=(3,))

    def run_attack(plane_u8, est, **kw):
        img4 = np.repeat(plane_u8[..., None], 4, axis=-1)        # what imread4_u8 returns for a gray PNG
        r = ref_ws.attack("mem", channels=(3,), pixel_estimator=est, imread=lambda f: img4, process_image=process, **kw)
        return np.float64(r["beta_hat"])

    # --- 64x64: linear predictors of the reference + an arbitrary host callable
    cov = formula.synthetic_images(3, 64, 64, seed=51)
    planes = [cov[0], formula.lsbr_embed(cov[1], 0.4, seed=3), formula.lsbr_embed(cov[2], 1.0, seed=4)]
    ests = {"KB": ref_filters.get_filter_estimator(filter_name="KB", flatten=False),
            "AVG": ref_filters.get_filter_estimator(filter_name="AVG", flatten=False),
            "cross": cross_mean}
    cfgs = [(w, cb) for w in (1, 0, -1) for cb in (False, True)]
    out["cfgs"] = np.array([[w, int(cb)] for w, cb in cfgs])
    for name, est in ests.items():
        out[f"beta64_{name}"] = np.array([[run_attack(p, est, weighted=w, correct_bias=cb) for w, cb in cfgs] for p in planes])
    out["beta64_KB_meanAVG9"] = np.array([run_attack(p, ests["KB"], weighted=1, mean_estimator=ref_ws.NAMED_FILTERS["AVG9"])
                                          for p in planes])
    out["filter64_KB"] = ests["KB"](process(np.repeat(planes[1][..., None], 4, axis=-1)))[..., 0]
    for k in ("KB", "AVG", "AVG9", "1"):
        out[f"named_{k}"] = ref_ws.NAMED_FILTERS[k]
        out[f"named2d_{k}"] = ref_filters.NAMED_FILTERS_2D[k]
    # --- 512x512: the UNet predictor (formula 'he' weights), x/255 -> net -> *255 -> crop as in unet/evaluate.py:45-51
    m = ref_model(2, "he"); m.input_dropout = None
    memo = {}

    def unet_est(x):
        key = x.tobytes()
        if key not in memo:
            with torch.no_grad():
                y = m(torch.from_numpy(np.ascontiguousarray((x / 255.).transpose(2, 0, 1)))[None])
            memo[key] = (y.numpy()[0, 0, 1:-1, 1:-1] * 255.)[..., None]
        return memo[key]

    c512 = formula.synthetic_images(1, 512, 512, seed=7)[0]
    p512 = [c512, formula.lsbr_embed(c512, 0.4, seed=5)]
    cfg512 = [(1, False), (0, False), (1, True), (0, True)]
    out["cfg512"] = np.array([[w, int(cb)] for w, cb in cfg512])
    out["beta512_unet"] = np.array([[run_attack(p, unet_est, weighted=w, correct_bias=cb) for w, cb in cfg512] for p in p512])


def main():
    sys.modules.setdefault("timm", types.ModuleType("timm"))
    losses = load_by_path("ref_losses", REF / "src" / "_defs" / "losses.py")
    metrics = load_by_path("ref_metrics", REF / "src" / "_defs" / "metrics.py")
    sys.path.insert(0, str(REF / "src"))
    import fabrika as fab
    groups = {
        "unet_fwd_small": lambda o: gen_forward_small(o),
        "unet_fwd_512": lambda o: gen_forward_512(o),
        "unet_grad": lambda o: gen_grads(o, losses),
        "dropout": lambda o: gen_dropout(o),
        "micro": lambda o: gen_micro(o),
        "losses": lambda o: gen_losses(o, losses, metrics),
        "adamw": lambda o: gen_adamw(o, losses),
        "ws_attack": lambda o: gen_ws_attack(o),
    }
    only = sys.argv[1:]
    for name, fn in groups.items():
        if only and name not in only:
            continue
        o = {}
        fn(o)
        np.savez_compressed(HERE / f"{name}.npz", **o)
        print(name, "->", sum(v.nbytes for v in o.values()) // 1024, "KiB raw,", len(o), "arrays")
    if not only or "fabrika" in only:
        gen_fabrika(fab)
    if not only or "png" in only:
        gen_png_kat()


if __name__ == "__main__":
    main()
