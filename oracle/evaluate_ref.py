"""Oracle for the per-image evaluate path (TEST INFRASTRUCTURE ONLY).

Restates, without torchvision / cv2 (neither is installed here):
  infere_single      src/unet/evaluate.py:31-52
  predict_unet       src/unet/evaluate.py:109-139
  get_timm_transform src/unet/data/loader.py:32-64  (ToTensor -> CenterCrop(512) -> Grayscale)
  Grayscale          src/_defs/loader.py:51-58      (1-ch passthrough, 4-ch -> channel 3)
  imread4_f32        src/_defs/imread.py:19-27      (gray PNG: all four planes equal the gray plane)

torchvision semantics restated:
  ToTensor on a float32 HxWxC ndarray = transpose to CxHxW, no rescale.
  CenterCrop(512): if a side is smaller, zero-pad it symmetrically first
  (left/top pad = (512 - s)//2, right/bottom = (512 - s + 1)//2), then crop with
  top = int(round((H - 512)/2.)) (Python round-half-even), same for left.
"""
import numpy as np
import torch


def center_crop(img: torch.Tensor, size: int = 512) -> torch.Tensor:
    c, h, w = img.shape
    if w < size or h < size:
        pl = (size - w) // 2 if w < size else 0
        pt = (size - h) // 2 if h < size else 0
        pr = (size - w + 1) // 2 if w < size else 0
        pb = (size - h + 1) // 2 if h < size else 0
        img = torch.nn.functional.pad(img, (pl, pr, pt, pb))
        c, h, w = img.shape
        if h == size and w == size:
            return img
    top = int(round((h - size) / 2.0))
    left = int(round((w - size) / 2.0))
    return img[:, top:top + size, left:left + size]


def transform_gray(x: np.ndarray) -> torch.Tensor:
    """(H,W,C) float32 -> (1,512,512) float32."""
    t = torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))
    t = center_crop(t, 512)
    if t.shape[0] == 4:
        t = t[3:]
    elif t.shape[0] != 1:
        raise NotImplementedError("colour -> gray conversion is outside the UNet path")
    return t


def infere_single(x: np.ndarray, model) -> np.ndarray:
    x_ = transform_gray((x / 255.).astype(np.float32))[None]
    y_ = model(x_)
    y = y_.detach().numpy()[0, 0, 1:-1, 1:-1] * 255.
    return y[..., None]


def predict_unet_array(x: np.ndarray, model) -> dict:
    """predict_unet after imread: x is (H,W,1) float32 with uint8 values."""
    x_hat = infere_single(x, model)
    x = x[1:-1, 1:-1]
    x_bar = (x.astype("uint8") ^ 1).astype("float32")
    beta_hat = np.mean((x - x_bar) * (x - x_hat))
    l1_hat = np.mean(np.abs(x - x_hat))
    return {"beta_hat": beta_hat, "l1": l1_hat}
