"""Torch-CPU restatement of the reference UNet (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/src/unet/model/unet.py:
  UniformDropout.forward   unet.py:32-42
  UNet.__init__            unet.py:54-135
  UNet.forward             unet.py:137-189
and src/unet/model/__init__.py:8-49 (get_model name parsing).

Written against torch.nn.functional only (the same ATen operators the reference
dispatches to), so it is both the arithmetic oracle (fp32, autograd backward)
and the timed CPU baseline of bench.py.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

_ENC = {1: ("e21", "e22"), 2: ("e31", "e32"), 3: ("e41", "e42"), 4: ("e51", "e52")}
# decoder level -> (upconv, first conv, second conv, skip tensor name); unet.py:112-132,168-186
_DEC = {4: ("upconv1", "d11", "d12", "xe42"), 3: ("upconv2", "d21", "d22", "xe32"),
        2: ("upconv3", "d31", "d32", "xe22"), 1: ("upconv4", "d41", "d42", "xe12")}

KB = torch.tensor([[[[-1., 2., -1.], [2., 0., 2.], [-1., 2., -1.]]]], dtype=torch.float32) / 4.  # unet.py:23-27


def parse_nsteps(name: str) -> int:
    """model/__init__.py:18-19."""
    if not name.lower().startswith("unet"):
        raise NotImplementedError(name)
    return int(name.split("_")[1])


def to_torch_state(sd: Dict[str, np.ndarray]) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v)).clone()) for k, v in sd.items())


def conv3x3_reflect(x, w, b):
    """nn.Conv2d(k=3, padding=1, padding_mode='reflect'): unet.py:73."""
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b)


def uniform_dropout(x: torch.Tensor, mask: torch.Tensor, channels=(0,)) -> torch.Tensor:
    """unet.py:32-42 with the Bernoulli keep-mask supplied by the caller
    (mask shape (N,1,H,W), 1 = keep).  Mutates ``x`` in place like the reference."""
    c = list(channels)
    m = mask.repeat((1, len(c), 1, 1))
    x_pad = F.pad(x[:, c], (1, 1, 1, 1), mode="reflect")
    x_kb = F.conv2d(x_pad, KB.repeat((1, len(c), 1, 1)))
    x[:, c] = x[:, c] * m + x_kb * (1 - m)
    return x


def unet_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], nsteps: int,
                 dropout_mask: Optional[torch.Tensor] = None, channels=(0,),
                 intermediates: Optional[dict] = None) -> torch.Tensor:
    """unet.py:137-189.  ``intermediates`` (if given) receives every named tensor."""
    t = {} if intermediates is None else intermediates
    if dropout_mask is not None:
        x = uniform_dropout(x, dropout_mask, channels)
    cv = lambda n, v: F.relu(conv3x3_reflect(v, sd[n + ".weight"], sd[n + ".bias"]))
    t["xe11"] = cv("e11", x)
    cur = t["xe12"] = cv("e12", t["xe11"])
    for lvl in range(1, nsteps + 1):
        a, b = _ENC[lvl]
        t[f"xp{lvl}"] = F.max_pool2d(cur, 2, 2)
        t["x" + a] = cv(a, t[f"xp{lvl}"])
        cur = t["x" + b] = cv(b, t["x" + a])
    for depth in range(nsteps, 0, -1):                  # block `depth` exists iff nsteps >= depth
        up, c1, c2, skip = _DEC[depth]
        xu = F.conv_transpose2d(cur, sd[up + ".weight"], sd[up + ".bias"], stride=2)
        t["xu" + up[-1]] = xu
        cat = torch.cat([xu, t[skip]], dim=1)           # upsampled first, skip second: unet.py:178,184
        t["x" + c1] = cv(c1, cat)
        cur = t["x" + c2] = cv(c2, t["x" + c1])
    z = F.conv2d(cur, sd["outconv.weight"], sd["outconv.bias"])
    t["logit"] = z
    return torch.sigmoid(z)


class UNetRef(torch.nn.Module):
    """Stock torch.nn restatement (same layers, same state_dict keys as unet.py:82-135).
    Used for autograd oracles and as the CPU baseline model."""

    def __init__(self, nsteps: int, in_channels: int = 1, out_channels: int = 1):
        super().__init__()
        self.nsteps = nsteps
        kw = dict(kernel_size=3, padding=1, padding_mode="reflect")
        C = torch.nn.Conv2d
        T = lambda a, b: torch.nn.ConvTranspose2d(a, b, kernel_size=2, stride=2)
        self.e11 = C(in_channels, 64, **kw); self.e12 = C(64, 64, **kw)
        if nsteps >= 1:
            self.e21 = C(64, 128, **kw); self.e22 = C(128, 128, **kw)
        if nsteps >= 2:
            self.e31 = C(128, 256, **kw); self.e32 = C(256, 256, **kw)
        if nsteps >= 3:
            self.e41 = C(256, 512, **kw); self.e42 = C(512, 512, **kw)
        if nsteps >= 4:
            self.e51 = C(512, 1024, **kw); self.e52 = C(1024, 1024, **kw)
        if nsteps >= 4:
            self.upconv1 = T(1024, 512); self.d11 = C(1024, 512, **kw); self.d12 = C(512, 512, **kw)
        if nsteps >= 3:
            self.upconv2 = T(512, 256); self.d21 = C(512, 256, **kw); self.d22 = C(256, 256, **kw)
        if nsteps >= 2:
            self.upconv3 = T(256, 128); self.d31 = C(256, 128, **kw); self.d32 = C(128, 128, **kw)
        if nsteps >= 1:
            self.upconv4 = T(128, 64); self.d41 = C(128, 64, **kw); self.d42 = C(64, 64, **kw)
        self.outconv = C(64, out_channels, kernel_size=1)

    def forward(self, x, dropout_mask=None):
        sd = dict(self.named_parameters())
        return unet_forward(x, sd, self.nsteps, dropout_mask)


def build_ref(nsteps: int, sd_np: Dict[str, np.ndarray]) -> UNetRef:
    m = UNetRef(nsteps, in_channels=sd_np["e11.weight"].shape[1], out_channels=sd_np["outconv.weight"].shape[0])
    m.load_state_dict(to_torch_state(sd_np))
    return m
