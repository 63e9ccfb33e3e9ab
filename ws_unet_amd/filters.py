"""Linear pixel predictors and plane selectors the WS estimator is configured with.

Mirrors the pieces of the reference that `src/ws/estimate.py:149-205 run` pulls in:
  NAMED_FILTERS_2D, get_coefficients, get_filter_estimator, infere_single   src/filters/evaluate.py:22-50,118-146
  get_processor_2d                                                          src/_defs/filters.py:72-83
The rest of the reference's filter tooling (OLS fits, HILL-cost weighted MAE tables) is outside the UNet path.
`infere_single` runs on the GPU (wsu_filter3x3_valid_f32); inside `ws.estimate` a `FilterEstimator` is recognised and its
taps are evaluated in the statistic kernel itself, so the prediction never exists in memory.
"""
import typing

import numpy as np

NAMED_FILTERS = {
    "KB": np.array([[-1], [+2], [-1], [+2], [-1], [+2], [-1], [+2]], dtype="float64") / 4.,
    "AVG": np.ones((8, 1)) / 8.,
}


def _k2d(rows, div):
    return np.array([rows], dtype="float32").T / div          # (3,3,1), [a][b][0] = rows[b][a] like the reference's `.T`


NAMED_FILTERS_2D = {
    "KB": _k2d([[-1, +2, -1], [+2, 0, +2], [-1, +2, -1]], 4.),
    "AVG": _k2d([[1, 1, 1], [1, 0, 1], [1, 1, 1]], 8.),
    "AVG9": _k2d([[1, 1, 1], [1, 1, 1], [1, 1, 1]], 9.),
    "1": _k2d([[0, 0, 0], [0, 1, 0], [0, 0, 0]], 1.),
}


def get_coefficients(filter_name: str, flatten: bool = True) -> np.ndarray:
    return NAMED_FILTERS[filter_name] if flatten else NAMED_FILTERS_2D[filter_name]


def infere_single(x: np.ndarray, model: np.ndarray) -> np.ndarray:
    """(H,W,C) float -> (H-2,W-2,1) float32: convolve(x / 255., model[..., ::-1], 'valid')[..., :1] * 255. for a
    single-channel 3x3 kernel (the only kind the UNet comparison uses)."""
    import torch
    from . import ops
    if model.ndim != 3 or model.shape != (3, 3, 1):
        raise NotImplementedError("only (3,3,1) kernels are on the GPU path")
    x0 = np.ascontiguousarray(np.asarray(x, dtype=np.float32)[..., 0])
    y = ops.filter3x3_valid(torch.from_numpy(x0)[None].cuda(), model[..., ::-1])
    return y[0].cpu().numpy()[..., None]


class FilterEstimator:
    """`lambda x: infere_single(x, kernel)` (filters/evaluate.py:144-146) as an object, so that callers can see the taps."""

    def __init__(self, kernel: np.ndarray):
        self.kernel = kernel

    def __call__(self, x: np.ndarray) -> np.ndarray:
        return infere_single(x, self.kernel)


def get_filter_estimator(*args, **kw) -> typing.Callable:
    return FilterEstimator(get_coefficients(*args, **kw))


def get_processor_2d(channels: typing.List[int]) -> typing.Callable:
    """Plane selector `x[..., channels].astype('float32')` (_defs/filters.py:72-83; the Bayer offsets are all None)."""
    channels = list(channels)

    def process_gray(x: np.ndarray) -> np.ndarray:
        return x[..., channels].astype("float32")

    process_gray.plane_selector = tuple(channels)          # lets the batched WS path skip the host arrays for the Y plane
    return process_gray
