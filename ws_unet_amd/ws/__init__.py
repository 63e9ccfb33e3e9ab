"""WS payload estimator built on the pixel predictor (reference src/ws/__init__.py:7; `roc` is out of scope)."""
from . import estimate  # noqa: F401
