"""Cover/stego pair loader for the UNet train step (`for inputs, (covers, alphas) in loader`).

The reference's training driver and dataset class for the UNet runs are not in the published tree, so the batch
composition is reconstructed from what IS there and from the run configs (models/unet/*/config.json):
  * rows come from `fabrika.cover_stego_spatial` over a split CSV (`tr_csv` / `va_csv`), one row per cover with its stego twin;
  * every pair contributes two samples, interleaved cover, stego as the detector's paired dataset does:
        (input = cover, target = cover, alpha = 0)   (input = stego, target = cover, alpha = row alpha)
    `covers_only=True` (the 'dropout' run) keeps only the first of the two;
  * the epoch order is a seeded permutation of the pairs (`reshuffle()` advances it); `shuffle=False` keeps fabrika's order.
That composition is this package's choice, not a pinned behaviour ("parity unpinned" for the sample order).

MI355X-side: files are decoded by libwsu_io on C++ threads into pinned buffers one batch ahead of the consumer, uploaded as
uint8 and scaled on the device (wsu_u8_to_unit_f32) -- 1 byte per pixel over PCIe instead of 4.  Data-parallel ranks take
disjoint, equally sized slices of every epoch (rank r gets pairs r, r+world, ...; the ragged tail is dropped so that all ranks
run the same number of steps and the per-step all-reduce never waits for a missing partner).
"""
from __future__ import annotations

import pathlib
import threading
import typing
from queue import Queue

import numpy as np
import torch

from .. import fabrika
from ..imread import read_luma_batch


@fabrika.cover_stego_spatial(iterator=None, convert_to=None, ignore_missing=True)
def _pair_rows(df, **kw):
    return df


class PairLoader:
    def __init__(self, dataset: typing.Union[str, pathlib.Path], split: typing.Optional[str], stego_method: typing.Optional[str],
                 alpha: typing.Optional[float], batch_size: int = 16, *, covers_only: bool = False, shuffle: bool = True,
                 seed: int = 0, rank: int = 0, world: int = 1, device: typing.Optional[torch.device] = None,
                 take_num_images: typing.Optional[int] = None, threads: typing.Optional[int] = None):
        per_pair = 1 if covers_only else 2
        if batch_size % per_pair:
            raise ValueError("batch_size must be even: every pair contributes a cover and a stego sample")
        self.dataset = pathlib.Path(dataset)
        df = _pair_rows(self.dataset, split=split, stego_method=stego_method, alpha=alpha, take_num_images=take_num_images)
        if not covers_only:
            df = df[~df["name_s"].isna()]
            if df.empty:
                raise ValueError(f"no cover/stego pairs for stego_method={stego_method!r} alpha={alpha!r} under {self.dataset}")
        self.covers = [str(n) for n in df["name_c"]]
        self.stegos = [] if covers_only else [str(n) for n in df["name_s"]]
        self.alphas = [0.0] * len(df) if covers_only else [float(a) for a in df["alpha_s"]]
        self.batch_size, self.per_pair, self.covers_only = batch_size, per_pair, covers_only
        self.shuffle, self.seed, self.epoch = shuffle, seed, 0
        self.rank, self.world, self.device, self.threads = rank, world, device, threads
        self._pinned = {}
        self._uploaded = {}                                             # slot -> event recorded behind its last upload

    # ---- epoch plan --------------------------------------------------------------------------------------
    def reshuffle(self) -> None:
        """Next epoch's permutation (the reference calls `tr_dataset.reshuffle()` before every epoch, detector/train.py:255)."""
        self.epoch += 1

    def pair_order(self) -> np.ndarray:
        n = len(self.covers)
        order = np.random.default_rng([self.seed, self.epoch]).permutation(n) if self.shuffle else np.arange(n)
        ppb = self.batch_size // self.per_pair                          # pairs per batch on one rank
        steps = n // (ppb * self.world)                                 # same on every rank; ragged tail dropped
        return order[:steps * ppb * self.world].reshape(steps, ppb, self.world)[:, :, self.rank]

    def __len__(self) -> int:
        return len(self.covers) // ((self.batch_size // self.per_pair) * self.world)

    # ---- one batch ---------------------------------------------------------------------------------------
    def _buffers(self, n, h, w, slot):
        key = (n, h, w, slot)
        if key not in self._pinned:
            pin = self.device is not None and torch.cuda.is_available()
            self._pinned[key] = torch.empty((n, h, w), dtype=torch.uint8, pin_memory=pin)
        return self._pinned[key]

    def _decode(self, pairs: np.ndarray, slot: int):
        files_in, files_cov, alphas = [], [], []
        for p in pairs:
            c = str(self.dataset / self.covers[p])
            files_in.append(c); files_cov.append(c); alphas.append(0.0)
            if not self.covers_only:
                files_in.append(str(self.dataset / self.stegos[p])); files_cov.append(c); alphas.append(self.alphas[p])
        from ..imread import png_shape, imread4_u8
        hw = png_shape(files_in[0]) or imread4_u8(files_in[0]).shape[:2]
        uniq = list(dict.fromkeys(files_in))                            # every cover is decoded once
        buf = self._buffers(len(uniq), hw[0], hw[1], slot)
        ev = self._uploaded.get(slot)
        if ev is not None:
            ev.synchronize()                                            # the upload that last read this pinned buffer has finished
        read_luma_batch(uniq, out=buf.numpy(), threads=self.threads)
        pos = {f: i for i, f in enumerate(uniq)}
        idx_in = torch.tensor([pos[f] for f in files_in]); idx_cov = torch.tensor([pos[f] for f in files_cov])
        return buf, idx_in, idx_cov, torch.tensor(alphas, dtype=torch.float32), slot

    def _finish(self, staged):
        buf, idx_in, idx_cov, alphas, slot = staged
        if self.device is None:                                         # host-logic mode: uint8 planes, no GPU involved
            return buf[idx_in].clone(), (buf[idx_cov].clone(), alphas)
        from .. import ops
        u8 = buf.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()                                                     # the producer waits for it before it reuses the slot
        self._uploaded[slot] = ev
        unit = ops.u8_to_unit(u8)[:, None]                              # (files,1,H,W) fp32 in [0,1], numpy's x / 255.
        return unit[idx_in.to(self.device)], (unit[idx_cov.to(self.device)], alphas.to(self.device))

    def __iter__(self):
        plan = self.pair_order()
        q: Queue = Queue(maxsize=1)

        def producer():
            try:
                for k, pairs in enumerate(plan):
                    q.put(("ok", self._decode(pairs, k % 3)))          # 3 slots: decoded ahead, queued, in use
                q.put(("end", None))
            except BaseException as e:                                  # surfaced in the consumer
                q.put(("err", e))

        t = threading.Thread(target=producer, daemon=True)
        t.start()
        while True:
            kind, item = q.get()
            if kind == "err":
                raise item
            if kind == "end":
                break
            yield self._finish(item)
        t.join()
