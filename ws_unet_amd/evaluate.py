"""Predictor / evaluate API of the reference (src/unet/evaluate.py:31-188) on the MI355X model.

Kept verbatim in name, arguments and result layout: `infere_single`, `get_model_name`, `predict_unet`,
`predict_unet_cover`, `predict_unet_stego`, `get_model_config`, `get_pretrained`.  Differences that a
drop-in user should know:
  * the model runs on the GPU; the `device` arguments are accepted for signature compatibility but the
    model's own device is used (there is no CPU path -- the reference hard-wires CPU, evaluate.py:27);
  * `infere_single` runs without building an autograd graph (the reference forgets no_grad, :48);
  * additions: `predict_unet_batch` + `predict_unet_cover_batched` / `predict_unet_stego_batched` evaluate
    whole batches on the device and bring back 8 bytes per image (WS statistic fused in
    wsu_ws_residual_stats), with the same rows / columns / order as the per-image functions.
"""
from __future__ import annotations

import glob
import json
import logging
import pathlib
import typing
from pathlib import Path

import numpy as np
import pandas as pd
import torch

from . import fabrika, ops
from .data import get_timm_transform
from .imread import imread4_f32, imread4_u8
from .model import get_model

DEVICE = torch.device("cuda")


def _model_device(model) -> torch.device:
    try:
        return next(model.parameters()).device
    except (StopIteration, AttributeError):
        return DEVICE


def range_fallback(model, collective: bool = False) -> bool:
    """One synchronising look at a planar model's range flag (UNet.range_exceeded: an activation beyond +-448 was stored since the last
    look -- the format's e4m3 residual saturates there and that value keeps only f16 accuracy, MAE ~1e-4, AT the gate).  If set: warn,
    switch the model to 'bf16x3s' (fp32-range storage) and return True -- the caller recomputes what it computed since the last look.
    The evaluate drivers call this once per data-set pass (one sync per pass), the per-image functions once per image (they synchronise
    on their result anyway).  `collective`: OR the flag over the ranks first (every rank of a sharded pass takes the same decision)."""
    planar = getattr(model, "mode", None) in ("f16f8p", "f16f8q", "f16f4p") and hasattr(model, "range_exceeded")
    if collective:
        # EVERY rank enters the all-reduce, whatever its local state (a rank that skipped it while the others entered would hang the job
        # or pair with a later collective).  A rank whose model already left the planar modes by a look of its OWN that the other ranks
        # have not seen yet (UNet.forward_features' first-forward look, a per-image call before the pass: `_range_switched` without
        # `_range_switch_synced`) contributes 1: the others follow and everybody recomputes, in the same arithmetic.
        from . import parallel
        if not hasattr(model, "_range_flag_tensor"):
            return False
        rf = model._range_flag_tensor(_model_device(model))
        unsynced = (not planar) and getattr(model, "_range_switched", False) and not getattr(model, "_range_switch_synced", False)
        hit = parallel.any_rank_flag(torch.ones_like(rf) if unsynced else rf)
        rf.zero_()
        if hit:
            model._range_switch_synced = True
        if not planar:
            return hit
    else:
        if not planar:
            return False
        hit = model.range_exceeded()
    if hit:
        logging.warning("ws_unet_amd.evaluate: activations beyond +-448 in mode '%s' (the planar format's e4m3 residual saturates there); "
                        "switching this model to mode 'bf16x3s' and recomputing", model.mode)
        model.mode = "bf16x3s"
        model._range_switched = True
    return hit


def infere_single(
    x: np.ndarray,
    model: typing.Callable,
    device=None,
) -> np.ndarray:
    """(H,W,1) float32 in 0..255  ->  (510,510,1) float32 prediction in 0..255 (evaluate.py:31-52)."""
    transform = get_timm_transform(
        mean=None, std=None, grayscale=True, demosaic_oracle=False, post_flip=False, post_rotate=False,
    )
    x_ = transform(x / 255.)[None].to(_model_device(model))
    with torch.no_grad():
        y_ = model(x_)
        if range_fallback(model):
            y_ = model(x_)
    y = y_.detach().cpu().numpy()[0, 0, 1:-1, 1:-1] * 255.
    return y[..., None]


def get_model_name(
    stego_method: str = "LSBR",
    model_dir: pathlib.Path = pathlib.Path("../models/unet"),
    device=None,
) -> str:
    """The single non-debug run under <model_dir>/<stego_method>/ whose config names `stego_method` and whose
    best checkpoint exists (evaluate.py:55-105).  RuntimeError unless exactly one matches."""
    found = []
    for cfg_file in map(pathlib.Path, glob.glob(str(pathlib.Path(model_dir) / stego_method / "*" / "config.json"))):
        run = cfg_file.parent.name
        with open(cfg_file) as f:
            config = json.load(f)
        try:
            ckpt = torch.load(cfg_file.parent / "model" / "best_model.pt.tar", map_location="cpu", weights_only=True)
        except FileNotFoundError:
            logging.warning(f"no model found for {run}, skipped")
            continue
        if config.get("debug", False):
            logging.warning(f"debug model {run} skipped")
            continue
        alpha = float(config["alpha"]) if config["alpha"] else config["alpha"]
        found.append({
            "model_name": run, "stego_method": config["stego_method"], "alpha": alpha, "loss": config["loss"],
            "network": config["network"], "drop_rate": config["drop_rate"], "epochs": ckpt["epoch"],
        })
    df = pd.DataFrame(found)
    if len(df):
        df = df[df.stego_method == stego_method]
    if len(df) < 1:
        raise RuntimeError(f"no model for {stego_method=} found")
    if len(df) > 1:
        raise RuntimeError(f"multiple models for {stego_method=} found")
    return df["model_name"].iloc[0]


def predict_unet(
    fname: str,
    model: torch.nn.Module,
    *,
    device=None,
    imread: typing.Callable = imread4_f32,
    **kw,
):
    """Per-image WS estimate and MAE (evaluate.py:109-139).  With this package's UNet and a 512x512 image the pixels go up as
    uint8 and only the two statistics come back (wsu_u8_to_unit_f32 -> forward -> wsu_ws_residual_stats, same float32 arithmetic);
    for any other predictor callable the reference's host formulas below are evaluated on its returned array."""
    if isinstance(model, torch.nn.Module) and hasattr(model, "forward_features") and imread is imread4_f32:
        # default reader + this package's UNet: the Y plane is decoded by libwsu_io (zlib + PNG unfilter + cv2's luma in C++, straight into a
        # pinned buffer) instead of PIL -- the same plane `imread4_f32(fname)[..., 3]` holds (tests/test_host_logic.py), 2.5x less host
        # time per image; files it does not support fall back to PIL inside read_luma_batch
        from .imread import png_shape
        now = _file_stamp(str(fname))                        # ONE stat per call: what is taken from the caches below must be of this file as it is now
        hit = _take_result(str(fname), model, now)           # computed in an earlier row's launch (the rows announced ahead ride along, below)
        if hit is not None:
            return {**kw, "beta_hat": hit[0], "l1": hit[1]}
        # rows announced ahead whose files are already decoded join this row's launch (up to _MICRO_BATCH images), and up to two further launches of
        # such rows are queued behind it before this call blocks on its own result: the reference's loop is one image per forward and one
        # blocking read-back per image (evaluate.py:48); rows, order and numbers stay those of that loop -- an image's statistics do not depend
        # on what else is in its batch -- but the GPU sees launches it can fill and works on the next one while Python hands out this one's rows
        h = _inflight_of(str(fname), model, now)
        if h is None:
            planes = _take_ahead(str(fname), now)            # decoded ahead by the iterator's lookahead (predict_unet_cover / _stego), if still valid
            if planes is None:
                planes = load_planes_u8([fname]) if png_shape(str(fname)) == (512, 512) else None
            if planes is not None:
                h = _submit_rows([(str(fname), planes, now)] + _take_ready_ahead(_MICRO_BATCH - 1), model)
        if h is not None:
            while len(_AHEAD["inflight"]) < _QUEUE_DEPTH and _ready_ahead() >= max(1, _MICRO_BATCH // 2):      # (a launch of one or two rows costs the host what a full one does)
                more = _take_ready_ahead(_MICRO_BATCH)
                if not more:
                    break
                _submit_rows(more, model)
            while True:                                      # collect in submission order up to this row's launch; the other rows' results wait for their calls
                g = _AHEAD["inflight"].pop(0)
                if g["model"] != id(model):                  # queued for another model (a caller alternating models outside a fabrika pass): not ours
                    continue
                beta, l1 = _collect_rows(g, model)
                for k, path in enumerate(g["paths"]):
                    if g is h and path == str(fname):
                        mine = (np.float32(beta[k]), np.float32(l1[k]))
                    else:
                        _AHEAD["results"][path] = (np.float32(beta[k]), np.float32(l1[k]), g["stamps"][k], id(model), getattr(model, "mode", None))
                if g is h:
                    break
            return {**kw, "beta_hat": mine[0], "l1": mine[1]}
    x = imread(fname)[..., 3:]
    if isinstance(model, torch.nn.Module) and hasattr(model, "forward_features") and x.shape[:2] == (512, 512):
        xi = np.ascontiguousarray(x[..., 0])
        if xi.dtype == np.uint8 or np.array_equal(xi, np.rint(xi)):
            x_u8 = torch.from_numpy(xi.astype(np.uint8))[None].to(_model_device(model))
            beta, l1, tripped = predict_u8_one_readback(x_u8, model)
            if tripped and range_fallback(model):
                beta, l1, _ = predict_u8_one_readback(x_u8, model)
            return {**kw, "beta_hat": np.float32(beta[0]), "l1": np.float32(l1[0])}
    x_hat = infere_single(x, model=model, device=device)
    x = x[1:-1, 1:-1]
    x_bar = (x.astype("uint8") ^ 1).astype("float32")          # integer LSB flip
    beta_hat = np.mean((x - x_bar) * (x - x_hat))
    l1_hat = np.mean(np.abs(x - x_hat))
    return {**kw, "beta_hat": beta_hat, "l1": l1_hat}


# ---- files ahead for the per-image API: while predict_unet works on row i (upload, forward, two scalars back: ~0.6 ms), helper threads decode
# the files of rows i + 1 .. i + 48 into the pinned ring -- one decode (~1.5 ms) is longer than everything else of a row (reference: serial,
# evaluate.py:142-149) -- and the rows already decoded when row i is asked for ride along in ITS launch (micro-batch), their results kept for
# their own calls: the per-image loop's GPU work becomes a few batch-8..16 forwards instead of one batch-1 forward and one blocking read-back per image
_MICRO_BATCH = max(1, int(__import__("os").environ.get("WSU_PER_IMAGE_BATCH", "16")))     # at most this many images in one per-image-API launch: the row asked for + decoded rows ahead
_QUEUE_DEPTH = 3                                             # launches in flight: the one a call waits for + two behind it (the GPU works while Python hands out rows)
_AHEAD_DEPTH = _QUEUE_DEPTH * _MICRO_BATCH                   # rows announced ahead (fabrika's python iterator, fn.lookahead_depth): the rows of the NEXT launches
                                                             # are announced while the rows of this one return from the cache, and decode during this launch
_AHEAD = {"pool": None, "pending": {}, "results": {}, "inflight": []}


def _file_stamp(path: str):
    import os
    try:
        st = os.stat(path)
        return (st.st_mtime_ns, st.st_size)
    except OSError:
        return None


def _decode_ahead(path: str):
    """-> (planes, hand-out number of the pinned buffer) or None"""
    from .imread import png_shape
    if png_shape(path) != (512, 512):
        return None
    planes = load_planes_u8([path])
    return None if planes is None else (planes, getattr(planes, "_wsu_issue", None))


def _ring_valid(res):
    """the pinned buffer of a finished ahead-decode still holds that decode (it has not been handed out again since)"""
    return res is not None and getattr(res[0], "_wsu_issue", None) == res[1]


def _lookahead(fname) -> None:
    if _AHEAD["pool"] is None:
        from concurrent.futures import ThreadPoolExecutor
        from ._io import usable_cores
        _AHEAD["pool"] = ThreadPoolExecutor(max_workers=max(4, min(_AHEAD_DEPTH, usable_cores())))
    pend = _AHEAD["pending"]
    while len(pend) > _AHEAD_DEPTH:                          # rows that were announced and never asked for
        pend.pop(next(iter(pend)))[0].cancel()
    # an entry = (future, file stamp): a decode lives in one buffer of load_planes_u8's pinned ring, which is handed out again after _NBUF1
    # further decodes -- a decode whose buffer was re-issued (_ring_valid), or of a file rewritten since, is dropped instead of uploaded
    pend[str(fname)] = (_AHEAD["pool"].submit(_decode_ahead, str(fname)), _file_stamp(str(fname)))


def _lookahead_reset() -> None:
    """Forget every announced-but-unconsumed decode and every computed-ahead result (start and end of a fabrika pass; a pass that raised midway
    leaves entries behind)."""
    pend = _AHEAD["pending"]
    while pend:
        pend.pop(next(iter(pend)))[0].cancel()
    _AHEAD["results"].clear()
    del _AHEAD["inflight"][:]                                # (queued launches of an abandoned pass simply finish; nobody reads them)


def _ready_ahead() -> int:
    """how many announced rows, oldest first, have finished decoding"""
    k = 0
    for fut, _ in _AHEAD["pending"].values():
        if not fut.done():
            break
        k += 1
    return k


def _take_ready_ahead(limit: int):
    """Announced rows whose decode has FINISHED, oldest first, at most `limit`: [(path, planes, stamp)].  Stops at the first row still decoding
    (nothing waits here); rows whose ring slot may have been re-issued, whose file changed, or that are not 512x512 are dropped -- their own call
    decodes them again."""
    out = []
    pend = _AHEAD["pending"]
    for path in list(pend):
        if len(out) >= limit:
            break
        fut, stamp = pend[path]
        if not fut.done():
            break
        pend.pop(path)
        if fut.cancelled():                                  # (a file rewritten since it was announced is caught when its row takes the result: the
            continue                                         # result carries the announce-time stamp -- no stat per candidate here)
        try:
            res = fut.result()
        except Exception:                                    # unreadable file: its own row raises the error, in order
            continue
        if _ring_valid(res):
            out.append((path, res[0], stamp))
    return out


def _submit_rows(rows, model):
    """rows [(path, pinned planes (1,H,W), stamp)] -> upload, forward, statistics and the copy of (beta_hat[n], l1[n], range flag) into a pinned
    host buffer, all queued on the current stream, nothing waits.  The handle joins _AHEAD['inflight']."""
    dev = _model_device(model)
    n = len(rows)
    if n == 1:
        x_u8 = rows[0][1].to(dev, non_blocking=True)
        mark_uploaded(rows[0][1])
    else:
        x_u8 = torch.empty((n,) + tuple(rows[0][1].shape[1:]), dtype=torch.uint8, device=dev)
        for k, (_, pl, _) in enumerate(rows):
            x_u8[k].copy_(pl[0], non_blocking=True)
        mark_uploaded([r[1] for r in rows])                  # one event behind the n uploads
    beta, l1 = predict_u8_batch(x_u8, model)
    rf = getattr(model, "_range_flag", None)
    planar = rf is not None and getattr(model, "mode", None) in ("f16f8p", "f16f8q", "f16f4p")
    parts = [beta.reshape(-1), l1.reshape(-1)] + ([rf.reshape(-1).view(torch.float32)] if planar else [])
    dev_v = torch.cat(parts)
    host = torch.empty(dev_v.shape, dtype=torch.float32, pin_memory=dev_v.is_cuda)
    host.copy_(dev_v, non_blocking=True)
    ev = None
    if dev_v.is_cuda:
        ev = torch.cuda.Event()
        ev.record()
    h = {"paths": [r[0] for r in rows], "stamps": [r[2] for r in rows], "x": x_u8, "host": host, "ev": ev, "n": n, "planar": planar,
         "model": id(model), "mode": getattr(model, "mode", None)}
    _AHEAD["inflight"].append(h)
    return h


def _collect_rows(h, model):
    """wait for ONE launch's results (its own event, not the stream): (beta_hat[n], l1[n]) as numpy.  A tripped range flag -- or a model that left
    the arithmetic this launch was computed in since (an earlier launch tripped it) -- recomputes the launch's images in the present arithmetic."""
    if h["ev"] is not None:
        h["ev"].synchronize()
    v = h["host"].numpy()
    n = h["n"]
    beta, l1 = v[:n].copy(), v[n:2 * n].copy()
    tripped = bool(h["planar"] and v[2 * n:].view(np.int32)[0] != 0)
    if (tripped and range_fallback(model)) or h["mode"] != getattr(model, "mode", None):
        beta, l1, _ = predict_u8_one_readback(h["x"], model)
    return beta, l1


def _inflight_of(path: str, model, now):
    """the queued launch that holds `path` (as the file is `now`) for this model, or None"""
    for h in _AHEAD["inflight"]:
        if h["model"] == id(model) and path in h["paths"] and h["stamps"][h["paths"].index(path)] == now:
            return h
    return None


def _take_result(path: str, model, now):
    """(beta_hat, l1) of `path` if an earlier launch of this pass already computed it with this model in its present arithmetic and the file is unchanged."""
    ent = _AHEAD["results"].pop(path, None)
    if ent is None or ent[3] != id(model) or ent[4] != getattr(model, "mode", None) or ent[2] != now:
        return None
    return ent[0], ent[1]


def _take_ahead(path: str, now=None):
    ent = _AHEAD["pending"].pop(path, None)
    if ent is None:
        return None
    fut, stamp = ent
    if stamp != (now if now is not None else _file_stamp(path)):                           # the file changed since it was announced: decode again
        fut.cancel()
        return None
    res = fut.result()
    return res[0] if _ring_valid(res) else None              # (a buffer handed out again since holds another file: decode again)


def _predict_unet_cover(*args, **kw):
    return predict_unet(*args, **kw)


def _predict_unet_stego(*args, **kw):
    return predict_unet(*args, **kw)


_predict_unet_cover.lookahead = _predict_unet_stego.lookahead = _lookahead
_predict_unet_cover.lookahead_reset = _predict_unet_stego.lookahead_reset = _lookahead_reset
_predict_unet_cover.lookahead_depth = _predict_unet_stego.lookahead_depth = _AHEAD_DEPTH
predict_unet_cover = fabrika.precovers(iterator="python", convert_to="pandas", ignore_missing=False, n_jobs=-1)(_predict_unet_cover)
predict_unet_stego = fabrika.stego_spatial(iterator="python", convert_to="pandas", ignore_missing=False, n_jobs=-1)(_predict_unet_stego)


# ---- batched device path ------------------------------------------------------------------------------

_POOL = None


def _decode_pool():
    """Thread pool for user-supplied `imread` callables (the default reader goes through libwsu_io instead)."""
    global _POOL
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        try:
            ncpu = len(os.sched_getaffinity(0))
        except AttributeError:
            ncpu = os.cpu_count() or 1
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(16, ncpu)))
    return _POOL


def predict_u8_batch(x_u8: torch.Tensor, model: torch.nn.Module):
    """x_u8: (N,H,W) uint8 on the model's device -> (beta_hat[N], l1[N]) fp32 device tensors.
    u8 -> /255 (wsu_u8_to_unit_f32) -> UNet forward -> WS residual statistics (wsu_ws_residual_stats)."""
    x01 = ops.u8_to_unit(x_u8)[:, None]
    with torch.no_grad():
        y = model(x01)
    return ops.ws_residual_stats(x_u8, y[:, 0].contiguous())


# ---- the per-image API with ONE read-back (round 4).  The reference's call pattern is one image per call
# (src/unet/evaluate.py:48,109-139).  Round 3 synchronised three times per image (the range flag, beta_hat, l1: three blocking 4-byte copies);
# now the three words leave the device as one 12-byte copy.  Measured on one MI355X box (profiles/r04/evaluate_loop.json.log): 1 079 -> 1 256
# images/s through predict_unet_cover.  (Also measured and NOT kept: the same chain as one hipGraph replay over static buffers -- 0.62 ms per
# call against 0.57 ms for the eagerly launched kernels, whose launches overlap the GPU's work on the previous ones.)
def predict_u8_one_readback(x_u8: torch.Tensor, model: torch.nn.Module):
    """predict_u8_batch + the model's range flag, brought to the host in ONE copy: (beta_hat[N], l1[N], flag_set) as numpy / bool."""
    beta, l1 = predict_u8_batch(x_u8, model)
    n = beta.numel()
    rf = getattr(model, "_range_flag", None)
    planar = rf is not None and getattr(model, "mode", None) in ("f16f8p", "f16f8q", "f16f4p")
    parts = [beta.reshape(-1), l1.reshape(-1)] + ([rf.reshape(-1).view(torch.float32)] if planar else [])
    v = torch.cat(parts).cpu().numpy()
    return v[:n], v[n:2 * n], bool(planar and v[2 * n:].view(np.int32)[0] != 0)


_PINNED = {}
_NBUF = 6                                                    # pinned buffers per chunk shape (decode of chunk k + 1 beside upload of chunk k ...)
_NBUF1 = _AHEAD_DEPTH + 2 * _MICRO_BATCH                     # ... and per single-image shape: more than the rows decoded ahead + the rows being uploaded
_PINNED_LOCK = __import__("threading").Lock()


# ---- pre-decoded uint8 shards (round 4, SURVEY 8d ".npy covers"): a data set's Y planes decoded ONCE into (N, H, W) uint8 .npy shards; an
# evaluate pass then copies rows out of a memory-mapped shard (~0.03 ms per image) instead of inflating a PNG (~2 ms per image and thread) -- the
# host budget of a file-fed pass stops scaling with the GPU rate.  Same bytes as the decode (tests/test_host_logic.py), same result table.
_U8_SHARDS = {"index": {}, "maps": {}}


def write_u8_shards(files, shard_dir, images_per_shard: int = 1024) -> pathlib.Path:
    """Decode `files` (absolute paths of equally sized images) into <shard_dir>/planes_%04d.npy + index.json (path -> shard, row, source file
    stamp).  Returns shard_dir.  Run once per data set; `use_u8_shards(shard_dir)` then serves load_planes_u8 from it."""
    from .imread import read_luma_batch
    shard_dir = pathlib.Path(shard_dir)
    shard_dir.mkdir(parents=True, exist_ok=True)
    files = [str(pathlib.Path(f).resolve()) for f in files]
    index = {}
    for k in range(0, len(files), images_per_shard):
        part = files[k:k + images_per_shard]
        planes = read_luma_batch(part)
        name = f"planes_{k // images_per_shard:04d}.npy"
        np.save(shard_dir / name, planes)
        for r, f in enumerate(part):
            index[f] = [name, r, list(_file_stamp(f) or (0, 0))]
    with open(shard_dir / "index.json", "w") as fh:
        json.dump({"format": "wsu-u8-shards-1", "files": index}, fh)
    return shard_dir


def use_u8_shards(shard_dir=None) -> int:
    """Serve load_planes_u8 from the shards under `shard_dir` (None: stop using shards).  Returns the number of indexed files."""
    _U8_SHARDS["index"], _U8_SHARDS["maps"] = {}, {}
    if shard_dir is None:
        return 0
    shard_dir = pathlib.Path(shard_dir)
    with open(shard_dir / "index.json") as fh:
        meta = json.load(fh)
    if meta.get("format") != "wsu-u8-shards-1":
        raise ValueError(f"{shard_dir}: not a wsu-u8-shards-1 index")
    _U8_SHARDS["index"] = {f: (str(shard_dir / name), row, tuple(stamp)) for f, (name, row, stamp) in meta["files"].items()}
    return len(_U8_SHARDS["index"])


def _planes_from_shards(fnames, out: np.ndarray) -> bool:
    """Fill out[i] with the pre-decoded plane of fnames[i]; False (out untouched or partly written: the caller decodes) unless EVERY file is
    indexed, unchanged on disk since it was decoded, and of the batch shape."""
    idx = _U8_SHARDS["index"]
    if not idx:
        return False
    ents = []
    for f in fnames:
        e = idx.get(f) or idx.get(str(pathlib.Path(f).resolve()))
        if e is None or e[2] != (_file_stamp(f) or (0, 0)):
            return False
        ents.append(e)
    for i, (path, row, _) in enumerate(ents):
        mm = _U8_SHARDS["maps"].get(path)
        if mm is None:
            mm = _U8_SHARDS["maps"][path] = np.load(path, mmap_mode="r")
        if mm.shape[1:] != out.shape[1:]:
            return False
        out[i] = mm[row]
    return True


def decode_budget(files, gpu_images_per_s: float = None, sample: int = 16) -> dict:
    """What a file-fed evaluate pass costs the HOST: the PNG decode time per image on ONE thread (measured on `sample`
    of the files, warm page cache), the decode threads a rank needs to keep its GPU fed at `gpu_images_per_s`, and what this process may use
    (cores it may run on / ranks on this node).  Logs a warning when the budget is short -- the pass is then host-bound:
    pre-decode the data set once (write_u8_shards / use_u8_shards) or give the ranks more cores."""
    import os
    import time
    from . import _io
    from .imread import read_luma_batch
    files = [str(f) for f in files[:sample]]
    res = {"decode_ms_per_image_per_thread": None, "threads_needed_per_rank": None, "usable_cores": _io.usable_cores(),
           "local_world_size": int(os.environ.get("LOCAL_WORLD_SIZE", "1")), "decode_threads_used": _io.default_threads()}
    if files:
        read_luma_batch(files[:2], threads=1)
        t0 = time.perf_counter()
        read_luma_batch(files, threads=1)
        res["decode_ms_per_image_per_thread"] = (time.perf_counter() - t0) / len(files) * 1e3
    if gpu_images_per_s and res["decode_ms_per_image_per_thread"]:
        res["threads_needed_per_rank"] = gpu_images_per_s * res["decode_ms_per_image_per_thread"] / 1e3
        have = res["usable_cores"] / max(1, res["local_world_size"])
        if have < res["threads_needed_per_rank"] and not _U8_SHARDS["index"]:
            logging.warning("ws_unet_amd.evaluate: this rank can decode on %.1f cores but needs %.1f decode threads to feed its GPU at %.0f images/s "
                            "(%.2f ms per PNG and thread): the pass is host-bound.  Pre-decode the data set (evaluate.write_u8_shards + use_u8_shards, "
                            "or --u8-shards) or run fewer ranks per host.", have, res["threads_needed_per_rank"], gpu_images_per_s, res["decode_ms_per_image_per_thread"])
    return res


def load_planes_u8(fnames, imread: typing.Callable = imread4_u8) -> typing.Optional[torch.Tensor]:
    """Y planes of a chunk of files as one (N,H,W) uint8 host tensor, or None when the files differ in shape.
    With the default reader the files are decoded by libwsu_io on C++ threads straight into a reused pinned buffer
    (PIL / cv2 style readers hold the GIL: Python threads do not scale them); any other `imread` is called per file."""
    fnames = [str(f) for f in fnames]
    if imread is imread4_u8:
        from .imread import png_shape, read_luma_batch
        hw = png_shape(fnames[0]) or imread4_u8(fnames[0]).shape[:2]
        key = (len(fnames), hw[0], hw[1])
        with _PINNED_LOCK:                                    # (the per-image lookahead decodes on a helper thread beside the caller's own reads)
            if key not in _PINNED:
                if len(_PINNED) > 8:
                    _PINNED.clear()
                pin = torch.cuda.is_available()
                nbuf = _NBUF1 if key[0] == 1 else _NBUF
                _PINNED[key] = {"bufs": [torch.empty(key, dtype=torch.uint8, pin_memory=pin) for _ in range(nbuf)], "next": 0,
                                "uploaded": [None] * nbuf}
            slot = _PINNED[key]
            i = slot["next"]                                 # a ring: chunk k+1 is decoded while chunk k is uploaded and chunk k-1 may still wait
            slot["next"] = (i + 1) % len(slot["bufs"])       # in the stream (submit / collect pipelining); the per-image API decodes 48 rows ahead
            slot["count"] = slot.get("count", 0) + 1
            slot["bufs"][i]._wsu_issue = slot["count"]       # which hand-out of this buffer the caller holds (_decode_ahead / _ring_valid)
            slot["bufs"][i]._wsu_ring = (slot, i)            # where mark_uploaded records the upload's event
        if slot["uploaded"][i] is not None:                  # the upload that last read this buffer (mark_uploaded)
            slot["uploaded"][i].synchronize()
            slot["uploaded"][i] = None
        buf = slot["bufs"][i]
        if _planes_from_shards(fnames, buf.numpy()):         # pre-decoded (use_u8_shards): a row copy per image instead of an inflate
            return buf
        try:
            read_luma_batch(fnames, out=buf.numpy())
        except ValueError:                                   # ragged shapes
            return None
        return buf
    imgs = list(_decode_pool().map(lambda f: np.ascontiguousarray(imread(f)[..., 3]), fnames))
    if len({im.shape for im in imgs}) != 1:
        return None
    return torch.from_numpy(np.stack(imgs))


def mark_uploaded(planes, event=None) -> None:
    """Record, for pinned buffer(s) handed out by load_planes_u8, the point in the current stream after which they may be overwritten
    (one event for all of them; `event`: an already recorded one)."""
    bufs = [planes] if isinstance(planes, torch.Tensor) else list(planes)
    rings = [getattr(b, "_wsu_ring", None) for b in bufs]
    if not any(r is not None for r in rings) or not torch.cuda.is_available():
        return
    if event is None:
        event = torch.cuda.Event()
        event.record()
    for r in rings:
        if r is not None:
            r[0]["uploaded"][r[1]] = event


def submit_unet_batch(fnames, *, model: torch.nn.Module, imread: typing.Callable = imread4_u8, prefetched=None):
    """First half of predict_unet_batch: upload + launch, nothing waits for the GPU.  Returns a handle for collect_unet_batch
    (ragged / non-512 chunks go through the per-image path right here and the handle carries their rows)."""
    planes = prefetched[0] if prefetched is not None else load_planes_u8(fnames, imread)
    if planes is None or tuple(planes.shape[1:]) != (512, 512):
        # CenterCrop(512) would change the geometry; only the per-image path defines what happens then
        res = [predict_unet(f, model, imread=imread4_f32) for f in fnames]
        return ("host", np.array([[r["beta_hat"], r["l1"]] for r in res], dtype=np.float32))
    x_u8 = planes.to(_model_device(model), non_blocking=True)
    mark_uploaded(planes)
    beta, l1 = predict_u8_batch(x_u8, model)
    return ("device", torch.stack([beta, l1], dim=1))


def collect_unet_batch(handle) -> np.ndarray:
    """Second half: (N, 2) float32 rows [beta_hat, l1] of a submitted chunk (waits for that chunk only)."""
    kind, val = handle
    return val if kind == "host" else val.cpu().numpy()


def predict_unet_batch(fnames, kws, *, model: torch.nn.Module, imread: typing.Callable = imread4_u8, device=None,
                       prefetched=None, **_ignored):
    """Batched predict_unet for `fabrika` iterator='batched': one result dict per (fname, kw).  `prefetched`: the planes of
    this chunk if the iterator already decoded them (load_planes_u8 run one chunk ahead)."""
    rows = collect_unet_batch(submit_unet_batch(fnames, model=model, imread=imread, prefetched=prefetched))
    return [{**kw, "beta_hat": rows[i, 0], "l1": rows[i, 1]} for i, kw in enumerate(kws)]


def pipelined_unet_rows(chunks, model: torch.nn.Module, imread: typing.Callable = imread4_u8):
    """(N_k, 2) rows per chunk of file names, in order, with the three stages of a chunk overlapped across chunks: decode of chunk
    k+1 on a helper thread, upload + GPU work of chunk k queued without waiting, read-back of chunk k-1."""
    from concurrent.futures import ThreadPoolExecutor
    chunks = list(chunks)
    if not chunks:
        return
    with ThreadPoolExecutor(max_workers=1) as ex:
        fut = ex.submit(load_planes_u8, chunks[0], imread)
        pending = None
        for k, chunk in enumerate(chunks):
            staged = (fut.result(),)
            fut = ex.submit(load_planes_u8, chunks[k + 1], imread) if k + 1 < len(chunks) else None
            handle = submit_unet_batch(chunk, model=model, imread=imread, prefetched=staged)
            if pending is not None:
                yield collect_unet_batch(pending)
            pending = handle
        yield collect_unet_batch(pending)


def _drop_model_kw(fn):
    def _clean(kws):
        return [{k: v for k, v in kw.items() if k not in ("model", "imread", "device")} for kw in kws]

    def wrapped(fnames, kws, prefetched=None):
        model = kws[0]["model"]
        extra = {k: kws[0][k] for k in ("imread",) if k in kws[0]}
        return fn(fnames, _clean(kws), model=model, prefetched=prefetched, **extra)

    # split form for the iterator's pipelining: submit(chunk k+1) is called before collect(chunk k)
    def submit(fnames, kws, prefetched=None):
        extra = {k: kws[0][k] for k in ("imread",) if k in kws[0]}
        return submit_unet_batch(fnames, model=kws[0]["model"], prefetched=prefetched, **extra), _clean(kws)

    def collect(handle):
        rows = collect_unet_batch(handle[0])
        return [{**kw, "beta_hat": rows[i, 0], "l1": rows[i, 1]} for i, kw in enumerate(handle[1])]

    if fn is predict_unet_batch:
        wrapped.submit, wrapped.collect = submit, collect
    # decode of the next chunk beside the GPU work of the current one (fabrika iterator='batched'); a 1-tuple so that
    # "ragged chunk" (None) stays distinguishable from "nothing prefetched"
    wrapped.prefetch = lambda fnames, kws: (load_planes_u8(fnames, kws[0].get("imread", imread4_u8)),)
    return wrapped


def _range_guarded(iterate):
    """A data-set pass of a batched driver, then ONE look at the model's range flag: if a planar forward of the pass left the format's
    full-accuracy range, the whole pass is recomputed in 'bf16x3s' (loudly).  No per-chunk synchronisation."""
    def run(dataset, *args, **kw):
        res = iterate(dataset, *args, **kw)
        model = kw.get("model")
        if model is not None and range_fallback(model):
            res = iterate(dataset, *args, **kw)
        return res
    run.__doc__ = iterate.__doc__
    return run


predict_unet_cover_batched = _range_guarded(fabrika.precovers(iterator="batched", convert_to="pandas", ignore_missing=False)(
    _drop_model_kw(predict_unet_batch)))
predict_unet_stego_batched = _range_guarded(fabrika.stego_spatial(iterator="batched", convert_to="pandas", ignore_missing=False)(
    _drop_model_kw(predict_unet_batch)))


def get_model_config(model_dir: pathlib.Path, stego_method: str, model_name: str) -> typing.Dict[str, typing.Any]:
    with open(pathlib.Path(model_dir) / stego_method / model_name / "config.json") as f:
        return json.load(f)


def get_pretrained(
    model_path,
    channels,
    *,
    model_name: str = None,
    device=None,
    mode: str = None,
):
    """Build the network named in <model_path>/<model_name>/config.json and load model/best_model.pt.tar
    (evaluate.py:162-188).  `channels` is accepted and ignored like in the reference."""
    model_path = Path(model_path)
    with open(model_path / model_name / "config.json") as f:
        config = json.load(f)
    dev = torch.device(device) if device is not None and torch.device(device).type == "cuda" else DEVICE
    model = get_model(config["network"], in_channels=1, out_channels=1, channel=[0], drop_rate=0., mode=mode).to(dev)
    checkpoint = torch.load(model_path / model_name / "model" / "best_model.pt.tar", map_location=dev, weights_only=True)
    model.load_state_dict(checkpoint["state_dict"])
    logging.info(f"model {model_name} loaded")
    return model


# ---- whole-dataset evaluate, batch-sharded over the ranks of a torch.distributed job -----------------------------

@fabrika.precovers(iterator=None, convert_to=None, ignore_missing=False)
def _cover_rows(df, **kw):
    return df


@fabrika.stego_spatial(iterator=None, convert_to=None, ignore_missing=False)
def _stego_rows(df, **kw):
    return df


def predict_unet_sharded(dataset, model: torch.nn.Module, *, stego_method: str = None, alpha: float = None, batch_size: int = 32,
                         **kw_iter) -> pd.DataFrame:
    """`predict_unet_cover` (stego_method None) / `predict_unet_stego` over a data set with the rows split contiguously over the
    ranks: every rank decodes and predicts its shard in batches, `(beta_hat, l1)` rows are all-gathered and every
    rank returns the full table in fabrika order with the per-image functions' columns.  Single process: same result, no
    collective."""
    from . import parallel
    dataset = pathlib.Path(dataset)
    if stego_method is None:
        df = _cover_rows(dataset, **kw_iter)
    else:
        kw = {"stego_method": stego_method, **({"alpha": alpha} if alpha is not None else {})}
        df = _stego_rows(dataset, **kw, **kw_iter)
    df = df.reset_index(drop=True)
    files = df["name"].tolist()                              # iterator=None hands over absolute paths (fabrika.py:104-110)

    def predict_shard(shard_files):                          # decode / GPU / read-back of consecutive chunks overlapped
        chunks = [shard_files[i:i + batch_size] for i in range(0, len(shard_files), batch_size)]
        rows = list(pipelined_unet_rows(chunks, model))
        return torch.from_numpy(np.concatenate(rows)) if rows else torch.zeros((0, 2), dtype=torch.float32)

    table = parallel.evaluate_sharded(files, predict_shard, None)
    if range_fallback(model, collective=True):                # one look per pass, OR-ed over the ranks: everybody recomputes together
        table = parallel.evaluate_sharded(files, predict_shard, None)
    table = table.cpu().numpy()
    df["name"] = [str(pathlib.Path(f).relative_to(dataset)) for f in files]
    df["beta_hat"], df["l1"] = table[:, 0], table[:, 1]
    if stego_method is not None:
        df = df.assign(stego_method=stego_method, **({"alpha": alpha} if alpha is not None else {}))
    return df


def main(argv=None) -> None:
    """The reference's `python unet/evaluate.py` (evaluate.py:190-233): covers + LSBR + HILLR stego rows of one trained model
    -> results/estimation/ws_<stego_method>.csv; run under torch.distributed.run to shard the rows over GPUs."""
    import argparse
    from . import parallel
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--data", default="../data")
    ap.add_argument("--model-dir", default="../models/unet")
    ap.add_argument("--stego-method", default="HILLR", help="which trained model: dropout | LSBR | HILLR")
    ap.add_argument("--eval-methods", nargs="*", default=["LSBR", "HILLR"])
    ap.add_argument("--out", default=None)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--mode", default=None)
    ap.add_argument("--u8-shards", default=None, help="directory of pre-decoded uint8 shards (written on first use by rank 0): the passes copy rows "
                                                       "out of memory-mapped .npy files instead of decoding PNGs")
    a = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    rank, world = parallel.init_from_env()
    model_dir = pathlib.Path(a.model_dir)
    model_name = get_model_name(model_dir=model_dir, stego_method=a.stego_method)
    model = get_pretrained(model_path=model_dir / a.stego_method, channels=(3,), model_name=model_name, mode=a.mode)
    if a.u8_shards:
        sd = pathlib.Path(a.u8_shards)
        if rank == 0 and not (sd / "index.json").exists():
            rows = [_cover_rows(pathlib.Path(a.data))] + [_stego_rows(pathlib.Path(a.data), stego_method=sm) for sm in a.eval_methods]
            write_u8_shards([f for df_ in rows for f in df_["name"].tolist()], sd)
        if world > 1:
            torch.distributed.barrier()
        logging.info("u8 shards: %d files indexed", use_u8_shards(sd))
    import time
    t0 = time.perf_counter()
    frames = [predict_unet_sharded(a.data, model, batch_size=a.batch_size)]
    dt = time.perf_counter() - t0
    if rank == 0:                                             # the host budget of a file-fed pass, stated once (no pass is repeated for it)
        nfiles = len(frames[0])
        rate = nfiles / world / dt if dt > 0 else None
        b = decode_budget([str(pathlib.Path(a.data) / n) for n in frames[0]["name"].tolist()], gpu_images_per_s=None)
        logging.info("evaluate: %d covers in %.2f s = %.0f images/s per rank end to end; PNG decode %.2f ms per image and thread, %d decode threads "
                     "per rank (usable cores %d / %d ranks on this node)%s", nfiles, dt, rate or 0.0, b["decode_ms_per_image_per_thread"] or 0.0,
                     b["decode_threads_used"], b["usable_cores"], b["local_world_size"], "; rows served from u8 shards" if a.u8_shards else "")
        if rate and b["decode_ms_per_image_per_thread"] and not a.u8_shards:
            busy = rate * b["decode_ms_per_image_per_thread"] / 1e3 / max(1, b["decode_threads_used"])
            if busy > 0.8:
                logging.warning("evaluate: the decode threads were ~%.0f %% busy at this rate: the pass is host-bound (see --u8-shards)", busy * 100)
    for sm in a.eval_methods:
        frames.append(predict_unet_sharded(a.data, model, stego_method=sm, batch_size=a.batch_size))
    df = pd.concat(frames)
    if rank == 0:
        out = pathlib.Path(a.out or f"../results/estimation/ws_{a.stego_method}.csv")
        out.parent.mkdir(parents=True, exist_ok=True)
        df.to_csv(out, index=False)
        logging.info(f"output saved to {out}")


if __name__ == "__main__":
    main()
