"""Oracle for the epoch meters of the UNet runs (TEST INFRASTRUCTURE ONLY).

What the reference computes (src/_defs/metrics.py):
  AverageMeter.update(val, n)   :35-38   running sum / count
  MAEMeter.update               :78-88   nanmean(|y_true - y_pred| * multiplier) of the batch, pushed with n=1
  WSMeter.update                :122-142 crop 1:-1, x255, flip LSB of round(x), per-image mean of
                                         (x - xbar)(x - xhat), clip at 0, mean |beta_hat - alpha/2|, pushed with n=1
Stated here as plain functions plus a running mean.
"""
import numpy as np


class RunningMean:
    def __init__(self):
        self.sum, self.count = 0.0, 0

    def push(self, value, n=1):
        self.sum += value * n
        self.count += n

    @property
    def avg(self):
        return self.sum / self.count


def mae_batch_value(y_true, y_pred, multiplier=1, mask=None, masked=None):
    if masked is True:
        y_true, y_pred = y_true[mask], y_pred[mask]
    elif masked is False:
        y_true, y_pred = y_true[~mask], y_pred[~mask]
    return np.nanmean(np.abs((y_true - y_pred) * multiplier))


def ws_batch_value(x, x_hat, alphas):
    xi = x[:, :, 1:-1, 1:-1] * 255.
    xh = x_hat[:, :, 1:-1, 1:-1] * 255.
    flipped = np.round(xi).astype("int") ^ 1
    per_pixel = (xi - flipped) * (xi - xh) / np.prod(xi.shape[1:])
    beta_hat = np.clip(per_pixel.sum(axis=(1, 2, 3)), 0, None)
    return np.mean(np.abs(beta_hat - alphas / 2.))
