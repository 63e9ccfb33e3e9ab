"""Weighted-stego (WS) payload estimator -- the caller of the UNet predictor (reference src/ws/estimate.py).

Same names and arguments: NAMED_FILTERS, attack (:55-136), attack_cover / attack_stego (:139-146), run (:149-205).
The statistic itself (local variance weights, LSB-flip residual product, clipping, bias correction) is ONE libwsu
kernel, wsu_ws_attack; there is no host implementation in this package.

  * `attack(fname, channels, pixel_estimator, ...)` is the per-image drop-in.  A `UNetEstimator` (what
    `get_unet_estimator` returns) keeps the prediction on the device; a `filters.FilterEstimator` is evaluated inside
    the kernel; any other callable is called on the host like in the reference and its (H-2,W-2,1) result uploaded.
  * `attack_batch` / `attack_cover_batched` / `attack_stego_batched` run whole batches (fabrika iterator='batched'):
    threaded PNG decode -> one u8 upload -> UNet forward(s) -> statistic -> 4 bytes per image back.
  * `run(..., batched=True)` is `run` on the batched iterators; the joblib iterators of the reference (:139,144) cannot
    carry a GPU model into worker processes, so the per-image decorators use iterator='python'.

Numerics: the reference's `scipy.signal.convolve` takes its FFT branch for a 512x512 plane, so its mu / mu2 carry float32
FFT noise (~1e-2 absolute on mu2); the kernel evaluates the nine taps directly (exact for the default AVG kernel on
uint8 pixels).  Agreement with the reference is therefore to that noise level (tests/golden/ws_attack.npz, rel 2e-4).
"""
from __future__ import annotations

import pathlib
import typing

import numpy as np
import torch

from .. import fabrika, filters, ops
from ..evaluate import _decode_pool, _model_device
from ..imread import imread4_u8

NAMED_FILTERS = filters.NAMED_FILTERS_2D


class UNetEstimator:
    """`predict(x (H,W,1) float32 0..255) -> (H-2,W-2,1)` closure of src/unet/__init__.py:110-121 as an object: callable
    like the reference's, and `.model` lets `attack` / `attack_batch` keep the prediction on the device."""

    def __init__(self, model):
        self.model = model

    def __call__(self, x: np.ndarray) -> np.ndarray:
        from ..evaluate import infere_single
        return infere_single(x, model=self.model)

    def __reduce__(self):
        # the reference ships its CPU predictor into joblib / loky workers (ws/estimate.py:139); a GPU model must not travel
        raise TypeError("UNetEstimator holds GPU state and cannot be pickled into worker processes: "
                        "use the fabrika iterators 'python' or 'batched' (ws.estimate.attack_cover / attack_cover_batched)")


def _as_u8_plane(x: np.ndarray) -> np.ndarray:
    """First channel of the processed image as uint8; the LSB flip is only defined for integer pixel values."""
    p = np.asarray(x)[..., 0]
    u = p.astype(np.uint8)
    if not np.array_equal(u.astype(p.dtype), p):
        raise ValueError("WS attack needs integer pixel values in 0..255")
    return np.ascontiguousarray(u)


def _device_of(pixel_estimator) -> torch.device:
    if isinstance(pixel_estimator, UNetEstimator):
        return _model_device(pixel_estimator.model)
    return torch.device("cuda")


def _unet_planes(model, x_u8: torch.Tensor, correct_bias: bool):
    """Full-frame network outputs in [0,1] for x and (if needed) for x_bar - x (estimate.py:89,127)."""
    with torch.no_grad():
        y = model(ops.u8_to_unit(x_u8)[:, None])[:, 0].contiguous()
        yb = model(ops.lsb_delta_unit(x_u8)[:, None])[:, 0].contiguous() if correct_bias else None
    return y, yb


def _stat(x_u8: torch.Tensor, pixel_estimator, mean_estimator, weighted, correct_bias, host_planes=None) -> torch.Tensor:
    """beta_hat[N] on the device for a batch of planes."""
    kw = dict(mean_filter=np.asarray(mean_estimator)[..., ::-1], weighted=int(weighted) if abs(int(weighted)) == 1 else 0,
              correct_bias=correct_bias)
    if isinstance(pixel_estimator, UNetEstimator):
        if x_u8.shape[1:] != (512, 512):
            raise ValueError("the UNet estimator works on 512x512 planes (CenterCrop(512) would change the geometry)")
        y, yb = _unet_planes(pixel_estimator.model, x_u8, correct_bias)
        return ops.ws_attack(x_u8, y, x_bias=yb, hat_scale=255.0, **kw)
    if isinstance(pixel_estimator, filters.FilterEstimator):
        return ops.ws_attack(x_u8, None, pixel_filter=np.asarray(pixel_estimator.kernel)[..., ::-1], **kw)
    # arbitrary host callable: reference call pattern, one image at a time
    hats, biases = [], []
    for xf in host_planes:
        h = np.asarray(pixel_estimator(xf), dtype=np.float32)
        if h.shape[:2] != (xf.shape[0] - 2, xf.shape[1] - 2):
            raise ValueError(f"pixel_estimator returned {h.shape} for an image of {xf.shape}")
        hats.append(h[..., 0])
        if correct_bias:
            xbar = (xf.astype(np.uint8) ^ 1).astype(np.float32)
            biases.append(np.asarray(pixel_estimator(xbar - xf), dtype=np.float32)[..., 0])
    x_hat = torch.from_numpy(np.stack(hats)).to(x_u8.device)
    x_bias = torch.from_numpy(np.stack(biases)).to(x_u8.device) if correct_bias else None
    return ops.ws_attack(x_u8, x_hat, x_bias=x_bias, hat_scale=1.0, **kw)


def attack(
    fname: str,
    channels: typing.List[int],
    pixel_estimator: typing.Union[np.ndarray, typing.Callable],
    mean_estimator: np.ndarray = NAMED_FILTERS["AVG"],
    correct_bias: bool = False,
    weighted: bool = 1,
    imread: typing.Callable = None,
    process_image: typing.Callable = None,
    **kw,
) -> dict:
    """WS estimate of one image (estimate.py:55-136): returns kw | {beta_hat, channels, weighted, correct_bias}."""
    x = process_image(imread(fname))                         # x_bar = process(x ^ 1) is formed on the device
    try:
        x_u8 = torch.from_numpy(_as_u8_plane(x))[None].to(_device_of(pixel_estimator))
        beta_hat = _stat(x_u8, pixel_estimator, mean_estimator, weighted, correct_bias, host_planes=[x])[0].item()
        beta_hat = np.float32(beta_hat)
    except ValueError:                                      # estimate.py:122-123
        beta_hat = None
    return kw | {
        "beta_hat": beta_hat,
        "channels": "".join(map(str, channels)),
        "weighted": weighted,
        "correct_bias": correct_bias,
    }


@fabrika.precovers(iterator="python", ignore_missing=True)
def attack_cover(*args, **kw):
    return attack(*args, **kw)


@fabrika.stego_spatial(iterator="python", ignore_missing=True)
def attack_stego(*args, **kw):
    return attack(*args, **kw)


# ---- batched device path ------------------------------------------------------------------------------

def _native_planes_ok(channels, pixel_estimator, imread, process_image) -> bool:
    """The default gray pipeline (Y plane, built-in predictor) needs no host arrays: native batched decode -> pinned buffer -> device."""
    builtin = isinstance(pixel_estimator, (UNetEstimator, filters.FilterEstimator))
    plain = process_image is None or getattr(process_image, "plane_selector", None) == (3,)
    return builtin and imread is imread4_u8 and plain and tuple(channels) == (3,)


def attack_batch(fnames, kws, *, channels, pixel_estimator, mean_estimator=NAMED_FILTERS["AVG"], correct_bias=False,
                 weighted=1, imread=imread4_u8, process_image=None, prefetched=None, **_ignored):
    """`attack` for a chunk of files (fabrika iterator='batched'): one result dict per (fname, kw)."""
    if _native_planes_ok(channels, pixel_estimator, imread, process_image):
        from ..evaluate import load_planes_u8
        u8 = prefetched[0] if prefetched is not None else load_planes_u8(fnames, imread)
        planes = None if u8 is None else [None] * len(fnames)
    else:
        process_image = process_image or filters.get_processor_2d(channels)
        planes = list(_decode_pool().map(lambda f: process_image(imread(f)), fnames))
        u8 = None
        if len({p.shape for p in planes}) != 1:
            planes = None
    if planes is None:
        process_image = process_image or filters.get_processor_2d(channels)
        return [attack(f, channels, pixel_estimator, mean_estimator, correct_bias, weighted, imread, process_image, **kw)
                for f, kw in zip(fnames, kws)]
    try:
        if u8 is None:
            u8 = torch.from_numpy(np.stack([_as_u8_plane(p) for p in planes]))
        x_u8 = u8.to(_device_of(pixel_estimator), non_blocking=True)
        beta = _stat(x_u8, pixel_estimator, mean_estimator, weighted, correct_bias, host_planes=planes).cpu().numpy()
    except ValueError:
        beta = [None] * len(fnames)
    tail = {"channels": "".join(map(str, channels)), "weighted": weighted, "correct_bias": correct_bias}
    return [kw | {"beta_hat": beta[i]} | tail for i, kw in enumerate(kws)]


_ATTACK_KEYS = ("channels", "pixel_estimator", "mean_estimator", "correct_bias", "weighted", "imread", "process_image")


def _split_attack_kw(fn):
    def wrapped(fnames, kws, prefetched=None):
        shared = {k: kws[0][k] for k in _ATTACK_KEYS if k in kws[0]}
        clean = [{k: v for k, v in kw.items() if k not in _ATTACK_KEYS} for kw in kws]
        return fn(fnames, clean, prefetched=prefetched, **shared)

    def prefetch(fnames, kws):
        k0 = kws[0]
        if not _native_planes_ok(k0["channels"], k0["pixel_estimator"], k0.get("imread", imread4_u8), k0.get("process_image")):
            return None
        from ..evaluate import load_planes_u8
        return (load_planes_u8(fnames, imread4_u8),)

    wrapped.prefetch = prefetch
    return wrapped


attack_cover_batched = fabrika.precovers(iterator="batched", ignore_missing=True)(_split_attack_kw(attack_batch))
attack_stego_batched = fabrika.stego_spatial(iterator="batched", ignore_missing=True)(_split_attack_kw(attack_batch))


def run(
    input_dir: pathlib.Path,
    stego_method: str,
    alpha: float,
    model_name: str,
    model_path: str,
    channels: typing.Tuple[int],
    imread: typing.Callable = imread4_u8,
    batched: bool = False,
    **kw,
):
    """WS attack over a data set with a named linear filter or a trained UNet as the pixel predictor (estimate.py:149-205)."""
    process_cover = filters.get_processor_2d(channels=channels)
    if model_name in NAMED_FILTERS:
        pixel_estimator = filters.get_filter_estimator(filter_name=model_name, flatten=False)
    else:
        from .. import get_unet_estimator
        pixel_estimator = get_unet_estimator(model_path=model_path, model_name=model_name, channels=channels)
        model_name = "UNet"
    if stego_method:
        fn = attack_stego_batched if batched else attack_stego
        kw_attack = {"stego_method": stego_method, "alpha": alpha}
    else:
        fn = attack_cover_batched if batched else attack_cover
        kw_attack = {}
    res = fn(
        input_dir,
        inbayer=None,
        **kw_attack,
        pixel_estimator=pixel_estimator,
        mean_estimator=NAMED_FILTERS["AVG"],
        model_name=model_name,
        channels=channels,
        process_image=process_cover,
        imread=imread,
        **kw,
    )
    res["channels"] = "".join(map(str, channels))
    res = res[~res.beta_hat.isna()]
    return res


def main(argv=None) -> None:
    """The reference's `python ws/estimate.py` (estimate.py:208-275): WS estimates of the covers and of the stego images at
    alpha 0.4 / 0.2 / 0.1 with the AVG and KB filters and with the trained UNets ('l1' = dropout run, 'l1ws' = the run trained on
    --train-method), one table -> results/estimation/ws_<train-method>.csv."""
    import argparse
    import pandas as pd
    from .. import get_model_name
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--data", default="../data/")
    ap.add_argument("--model-dir", default="../models/unet")
    ap.add_argument("--train-method", default="LSBR", help="stego method the l1ws UNet was trained on")
    ap.add_argument("--stego-methods", nargs="*", default=["LSBR"])
    ap.add_argument("--alphas", nargs="*", type=float, default=[.4, .2, .1])
    ap.add_argument("--filters", nargs="*", default=["AVG", "KB"])
    ap.add_argument("--losses", nargs="*", default=["l1", "l1ws"])
    ap.add_argument("--weighted", type=int, default=0)
    ap.add_argument("--correct-bias", action="store_true")
    ap.add_argument("--per-image", action="store_true", help="use the per-image iterators instead of the batched ones")
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)
    model_dir = pathlib.Path(a.model_dir)
    settings = [(None, .0)] + [(sm, al) for sm in a.stego_methods for al in a.alphas]
    common = dict(demosaic=None, channels=(3,), correct_bias=a.correct_bias, weighted=a.weighted, batched=not a.per_image)
    res = []
    for stego_method, alpha in settings:
        for model_name in a.filters:
            res.append(run(input_dir=pathlib.Path(a.data), stego_method=stego_method, alpha=alpha, model_path=None,
                           model_name=model_name, **common))
    for loss in a.losses:
        train_method = a.train_method if loss == "l1ws" else "dropout"
        name = get_model_name(stego_method=train_method, model_dir=model_dir)
        for stego_method, alpha in settings:
            r = run(input_dir=pathlib.Path(a.data), stego_method=stego_method, alpha=alpha, model_path=model_dir / train_method,
                    model_name=name, **common)
            r["model_name"] = f"UNet_{loss}" + (f"_{train_method}" if loss == "l1ws" else "")
            res.append(r)
    res = pd.concat(res).reset_index(drop=True)
    res["stego_method"] = res["stego_method"].fillna("Cover") if "stego_method" in res else "Cover"
    out = pathlib.Path(a.out or f"../results/estimation/ws_{a.train_method}.csv")
    out.parent.mkdir(parents=True, exist_ok=True)
    res.to_csv(out, index=False)
    print(f"output saved to {out}")


if __name__ == "__main__":
    main()
