"""Loss classes of the UNet runs with the reference's names and call signature
(src/_defs/losses.py: L1Loss :28-36, WSLoss :45-90, L1WSLoss :93-121):

    criterion(outputs, targets=(covers, alphas), inputs) -> scalar tensor

The value and dLoss/doutputs come from ONE fused libwsu call (wsu_l1ws_loss_fwd_bwd, K8): per-image
fp64 reductions, the integer LSB flip `round(x*255) ^ 1` done in integer arithmetic (bit-exact), the
gradient written in the same pass structure.  `loss.backward()` just scales that stored gradient.
L1WSLoss = L1 + WS unweighted (the configs' `loss_lambda` is unused by the reference, losses.py:114-116).
L2Loss (losses.py:39-42, unused by the published runs) rides the same kernel with the squared error in the L1 slot.
"""
import torch

from . import ops


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outputs, covers, inputs, alphas, use_l1, use_ws):
        loss, dout, parts, beta = ops.l1ws_loss_fwd_bwd(outputs.contiguous(), covers.contiguous(), inputs.contiguous(),
                                                        alphas, use_l1, use_ws)
        ctx.dout = dout
        ctx.mark_non_differentiable(parts, beta)
        return loss, parts, beta

    @staticmethod
    def backward(ctx, gl, _gp, _gb):
        return ctx.dout * gl, None, None, None, None, None


class _Base(torch.nn.Module):
    use_l1, use_ws = True, True

    def __init__(self, device=None):
        super().__init__()
        self.device = device
        self.last_parts = None          # (l1, ws) of the last call, device tensor
        self.last_beta_hat = None

    def forward(self, outputs, targets, inputs=None, *args, **kw):
        covers, alphas = targets
        if inputs is None:
            if self.use_ws:
                raise TypeError("WS loss needs the network inputs: criterion(outputs, (covers, alphas), inputs)")
            inputs = covers
        if alphas is None:
            alphas = torch.zeros(outputs.shape[0], device=outputs.device)
        alphas = torch.as_tensor(alphas, dtype=torch.float32, device=outputs.device)
        loss, self.last_parts, self.last_beta_hat = _FusedLoss.apply(outputs, covers, inputs, alphas, self.use_l1, self.use_ws)
        return loss

    def to(self, device, *args, **kw):
        self.device = device
        return self


class L1Loss(_Base):
    use_l1, use_ws = True, False


class L2Loss(_Base):
    use_l1, use_ws = 2, False


class WSLoss(_Base):
    use_l1, use_ws = False, True


class L1WSLoss(_Base):
    use_l1, use_ws = True, True


def get_loss(name: str):
    """Config key `loss` of models/unet/*/config.json: 'l1' (dropout run) or 'l1ws' (LSBR / HILLR runs)."""
    try:
        return {"l1": L1Loss, "l2": L2Loss, "ws": WSLoss, "l1ws": L1WSLoss}[name]()
    except KeyError:
        raise NotImplementedError(f"loss {name} not implemented")
