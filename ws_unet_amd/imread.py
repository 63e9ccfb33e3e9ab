"""Image readers with the reference's names (src/_defs/imread.py:11-27), on PIL only (cv2 is not a
dependency here).

imread4_u8 returns (H, W, 4) = [R, G, B, Y].  For the gray PNGs of the data set all four planes equal
the stored gray plane: the reference reads with cv2.imread (gray replicated to BGR) and cv2's BGR2GRAY,
whose fixed-point weights (R 4899, G 9617, B 1868, sum 2**14, rounding shift 14) map v,v,v -> v exactly
(SURVEY.md a7).  Colour inputs use the same fixed-point luma; that branch is restated from OpenCV's
documented coefficients and is not pinned by a reference fixture (cv2 absent in the build image).
"""
import numpy as np
from PIL import Image


def imread_u8(fname) -> np.ndarray:
    x = np.array(Image.open(fname))
    if x.ndim == 2:
        x = x[..., None]
    return x


def imread_f32(fname) -> np.ndarray:
    return imread_u8(fname).astype("float32")


def imread4_u8(fname) -> np.ndarray:
    img = Image.open(str(fname))
    if img.mode not in ("L", "RGB"):
        img = img.convert("RGB")
    x = np.array(img)
    if x.ndim == 2:
        return np.repeat(x[..., None], 4, axis=-1)
    rgb = x[..., :3].astype(np.int64)
    y = (rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + (1 << 13)) >> 14
    return np.concatenate([x[..., :3], y.astype(np.uint8)[..., None]], axis=-1)


def imread4_f32(fname) -> np.ndarray:
    return imread4_u8(fname).astype("float32")
