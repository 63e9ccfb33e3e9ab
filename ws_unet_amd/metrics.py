"""Epoch meters of the UNet runs with the reference's class names (src/_defs/metrics.py:19-142):
AverageMeter, LossMeter, MAEMeter, WSMeter.  Host-side numpy on a few scalars per batch; the tags they
feed are train|val/{loss, mae, ws} like the published tfevents."""
import numpy as np


class AverageMeter:
    """Running average: update(val, n) adds val*n to the sum (metrics.py:35-38)."""
    name = None

    def __init__(self):
        self.reset()

    def reset(self):
        self.avg, self.sum, self.count = 0, 0, 0

    def update(self, val, n=1):
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __str__(self):
        return f"{self.name} {self.avg:.3f}"


class LossMeter(AverageMeter):
    name = "loss"


class MAEMeter(AverageMeter):
    """Batch value = nanmean(|y_true - y_pred| * multiplier), each batch counted once (metrics.py:78-88)."""
    name = "mae"

    def __init__(self, multiplier: int = 1, masked: bool = None):
        super().__init__()
        self.multiplier, self.masked = multiplier, masked

    def update(self, y_true, y_pred, mask=None):
        if self.masked is True:
            y_true, y_pred = y_true[mask], y_pred[mask]
        elif self.masked is False:
            y_true, y_pred = y_true[~mask], y_pred[~mask]
        super().update(np.nanmean(np.abs((y_true - y_pred) * self.multiplier)))


class WSMeter(AverageMeter):
    """Batch value = mean_n |clip(beta_hat_n, 0) - alpha_n/2| with beta_hat on the [1:-1,1:-1] interior in 0..255
    units and the LSB flip of round(x) done on integers (metrics.py:122-142)."""
    name = "ws"

    def update(self, x, x_hat, alphas):
        xi = x[:, :, 1:-1, 1:-1] * 255.
        xh = x_hat[:, :, 1:-1, 1:-1] * 255.
        x_bar = np.round(xi).astype("int") ^ 1
        beta_hat = np.sum((xi - x_bar) * (xi - xh) / np.prod(xi.shape[1:]), axis=(1, 2, 3))
        beta_hat = np.clip(beta_hat, 0, None)
        super().update(np.mean(np.abs(beta_hat - alphas / 2.)))
