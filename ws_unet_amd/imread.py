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


def png_shape(fname):
    """(H, W) of a PNG without decoding pixels (wsu_png_shape); None if libwsu_io cannot parse the file."""
    import ctypes
    from . import _io
    h, w = ctypes.c_int(0), ctypes.c_int(0)
    rc = _io.load().wsu_png_shape(str(fname).encode(), ctypes.byref(h), ctypes.byref(w))
    return (h.value, w.value) if rc in (_io.PNG_OK,) else None


def read_luma_batch(fnames, out=None, threads=None):
    """Y planes `imread4_u8(f)[..., 3]` of a list of equally sized images as one (N,H,W) uint8 array, decoded on C++ threads
    (wsu_png_read_luma_batch, include/wsu_io.h).  `out`: optional destination (numpy array or anything exposing `.ctypes` /
    a writable buffer, e.g. a pinned torch tensor's `.numpy()`); its shape fixes (H, W), otherwise the first file does.
    Files the native reader does not support (palette, alpha, 16 bit, interlaced) are read with PIL; a shape mismatch or an
    unreadable file raises."""
    import ctypes
    from . import _io
    lib = _io.load()
    names = [str(f) for f in fnames]
    n = len(names)
    if out is None:
        hw = png_shape(names[0]) if n else (0, 0)
        if hw is None:
            hw = imread4_u8(names[0]).shape[:2]
        out = np.empty((n, hw[0], hw[1]), np.uint8)
    assert out.dtype == np.uint8 and out.ndim == 3 and out.shape[0] >= n and out.flags.c_contiguous
    if n == 0:
        return out[:0]
    h, w = out.shape[1], out.shape[2]
    arr = (ctypes.c_char_p * n)(*[s.encode() for s in names])
    status = (ctypes.c_int * n)()
    failed = lib.wsu_png_read_luma_batch(arr, n, out.ctypes.data, h, w, threads or _io.default_threads(), status)
    if failed < 0:
        raise _io.WsuIoError("wsu_png_read_luma_batch: bad arguments")
    if failed:
        for i, rc in enumerate(status):
            if rc == _io.PNG_OK:
                continue
            if rc == _io.PNG_UNSUPPORTED:
                y = imread4_u8(names[i])[..., 3]
                if y.shape != (h, w):
                    raise ValueError(f"{names[i]}: shape {y.shape} differs from the batch shape {(h, w)}")
                out[i] = y
            elif rc == _io.PNG_SHAPE:
                raise ValueError(f"{names[i]}: shape differs from the batch shape {(h, w)}")
            else:
                raise OSError(f"{names[i]}: cannot read PNG (libwsu_io code {rc})")
    return out[:n]
