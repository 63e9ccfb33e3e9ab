"""Input transform with the reference's name and arguments (src/unet/data/loader.py:32-64,
src/_defs/loader.py:51-103), without torchvision.

Pipeline for an (H, W, C) ndarray:  ToTensor -> CenterCrop(512) -> [Grayscale] -> [ParityOracle] ->
[DemosaicOracle] -> [Normalize] -> [random flips] -> [random rot90].

torchvision semantics restated (not importable here, so pinned only by the restatement in
oracle/evaluate_ref.py and the 512x512 identity case):
  ToTensor     HWC -> CHW; uint8 input is divided by 255, float input is left unscaled
  CenterCrop   sides shorter than the crop are zero-padded symmetrically first
               (left/top (s-d)//2, right/bottom (s-d+1)//2), then the crop starts at round((d-s)/2)
"""
from __future__ import annotations

import numpy as np
import torch


def to_tensor(x: np.ndarray) -> torch.Tensor:
    if x.ndim == 2:
        x = x[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))
    return t.float().div(255) if t.dtype == torch.uint8 else t


def center_crop(img: torch.Tensor, size: int = 512) -> torch.Tensor:
    _, h, w = img.shape
    if h < size or w < size:
        pl, pr = ((size - w) // 2, (size - w + 1) // 2) if w < size else (0, 0)
        pt, pb = ((size - h) // 2, (size - h + 1) // 2) if h < size else (0, 0)
        img = torch.nn.functional.pad(img, (pl, pr, pt, pb))
        _, h, w = img.shape
        if h == size and w == size:
            return img
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return img[:, top:top + size, left:left + size]


def grayscale(img: torch.Tensor) -> torch.Tensor:
    """_defs/loader.py:51-58: 1 channel passes through, 4 channels -> channel 3 (the Y plane of imread4),
    3 channels -> ITU-R 601-2 luma like torchvision's Grayscale."""
    if img.shape[0] == 1:
        return img
    if img.shape[0] == 4:
        return img[3:]
    r, g, b = img[0], img[1], img[2]
    return (0.2989 * r + 0.587 * g + 0.114 * b).to(img.dtype)[None]


def parity_oracle(img: torch.Tensor) -> torch.Tensor:
    """_defs/loader.py:74-84: append the LSB plane of round(img*255)."""
    return torch.cat([img, (torch.round(img * 255).int() & 1).to(img.dtype)], dim=0)


def lsbr_reference(img: torch.Tensor) -> torch.Tensor:
    """_defs/loader.py:61-71: append the image with its LSB cleared."""
    return torch.cat([img, (torch.round(img * 255).int() & ~1) / 255.], dim=0)


def demosaic_oracle(img: torch.Tensor) -> torch.Tensor:
    """_defs/loader.py:87-103: append the three Bayer (RGGB) indicator planes."""
    grid = torch.zeros(3, *img.shape[1:], dtype=img.dtype)
    grid[0, ::2, ::2] = 1
    grid[1, 1::2, ::2] = 1
    grid[1, ::2, 1::2] = 1
    grid[2, 1::2, 1::2] = 1
    return torch.cat([img, grid], dim=0)


class Compose:
    def __init__(self, steps):
        self.steps = steps

    def __call__(self, x):
        for s in self.steps:
            x = s(x)
        return x


def get_timm_transform(
    mean: float,
    std: float,
    grayscale: bool = False,
    parity_oracle: bool = False,
    demosaic_oracle: bool = False,
    post_flip: bool = False,
    post_rotate: bool = False,
):
    g = globals()
    steps = [to_tensor, lambda t: center_crop(t, 512)]
    if grayscale:
        steps.append(g["grayscale"])
    if parity_oracle:
        steps.append(g["parity_oracle"])
    if demosaic_oracle:
        steps.append(g["demosaic_oracle"])
    if mean is not None and std is not None:
        m = torch.as_tensor(mean, dtype=torch.float32).reshape(-1, 1, 1)
        s = torch.as_tensor(std, dtype=torch.float32).reshape(-1, 1, 1)
        steps.append(lambda t: (t - m) / s)
    if post_flip:
        steps.append(lambda t: t.flip(-1) if torch.rand(1).item() < 0.5 else t)
        steps.append(lambda t: t.flip(-2) if torch.rand(1).item() < 0.5 else t)
    if post_rotate:
        steps.append(lambda t: torch.rot90(t, int(torch.randint(0, 4, (1,)).item()), dims=(-2, -1)))
    return Compose(steps)
