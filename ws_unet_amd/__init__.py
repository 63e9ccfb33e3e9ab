"""ws_unet_amd -- MI355X-native UNet pixel predictor for WS steganalysis.

Package facade mirroring the reference's `src/unet/__init__.py:9-11,110-121`:
    from ws_unet_amd import get_model, load_model, infere_single, get_model_name, get_model_config,
                            get_pretrained, get_unet_estimator
Everything heavy (torch, the HIP library) is imported lazily on first attribute access so that
`import ws_unet_amd.formula` stays cheap.
"""
import importlib

__all__ = ["get_model", "load_model", "infere_single", "get_model_name", "get_model_config",
           "get_pretrained", "get_unet_estimator", "data"]

_LAZY = {
    "get_model": ("ws_unet_amd.model", "get_model"),
    "load_model": ("ws_unet_amd.model", "load_model"),
    "infere_single": ("ws_unet_amd.evaluate", "infere_single"),
    "get_model_name": ("ws_unet_amd.evaluate", "get_model_name"),
    "get_model_config": ("ws_unet_amd.evaluate", "get_model_config"),
    "get_pretrained": ("ws_unet_amd.evaluate", "get_pretrained"),
}


def __getattr__(name):
    if name in _LAZY:
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(mod), attr)
    if name in ("data", "evaluate", "model", "fabrika", "ops", "formula", "losses", "metrics", "imread", "parallel", "trainer", "filters", "ws"):
        return importlib.import_module(f"ws_unet_amd.{name}")
    raise AttributeError(name)


def get_unet_estimator(*args, **kw):
    """`predict(x: (H,W,1) float32 0..255) -> (H-2,W-2,1)` over a pretrained model (reference
    src/unet/__init__.py:110-121), returned as a callable `ws.estimate.UNetEstimator` whose `.model` lets the WS
    estimator keep predictions on the device.  It holds GPU state: it cannot be pickled into joblib/loky workers
    (the reference's ws/estimate.py:139 does that with its CPU model) -- use fabrika iterator='python' or 'batched'."""
    from .evaluate import get_pretrained
    from .ws.estimate import UNetEstimator
    return UNetEstimator(get_pretrained(*args, **kw))
