"""Restatement of the reference loss classes (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/src/_defs/losses.py:
  L1Loss    :28-36   mean |covers - outputs|
  L2Loss    :39-42
  WSLoss    :45-90   relu(sum_n w (in255 - flip(in255)) (in255 - out255)), mean_n |beta_hat - alpha/2|
  L1WSLoss  :93-121  L1 + WS, unweighted (loss_lambda unused, :114-116)
Signature kept: forward(outputs, targets=(covers, alphas), inputs).
"""
import torch


def l1_loss(outputs, targets, *args, **kw):
    covers, _ = targets
    return torch.mean(torch.abs(covers - outputs))


def l2_loss(outputs, targets, *args, **kw):
    covers, _ = targets
    return torch.mean((covers - outputs) ** 2)


def ws_error(outputs, inputs, betas):
    inputs = inputs * 255.
    outputs = outputs * 255.
    inputs_bar = (torch.round(inputs).int() ^ 1).float()
    weights = torch.ones_like(inputs) / (torch.numel(inputs) / float(inputs.size(0)))
    betas_hat = torch.sum(weights * (inputs - inputs_bar) * (inputs - outputs), dim=(1, 2, 3))
    betas_hat = torch.nn.functional.relu(betas_hat)
    return torch.abs(betas_hat - betas)


def ws_loss(outputs, targets, inputs):
    _, alphas = targets
    return torch.mean(ws_error(outputs, inputs, alphas / 2.))


def l1ws_loss(outputs, targets, inputs):
    return l1_loss(outputs, targets) + ws_loss(outputs, targets, inputs)


LOSSES = {"l1": l1_loss, "l2": l2_loss, "ws": ws_loss, "l1ws": l1ws_loss}
