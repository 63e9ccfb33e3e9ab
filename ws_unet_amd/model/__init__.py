"""Model factory with the reference's interface (src/unet/model/__init__.py:8-62)."""
import pathlib

import torch

from . import unet
from .unet import UNet, UniformDropout  # noqa: F401


def get_model(
    name: str,
    in_channels: int,
    out_channels: int = 1,
    channel=[0],
    drop_rate: float = 0.,
    *,
    mode: str = None,
) -> torch.nn.Module:
    """``name`` = 'unet_<nsteps>' (reference :18-27).  The reference's 'cnn' branch (:30-43) points at a class
    that does not exist in its tree, so any other name raises NotImplementedError like the reference's
    fall-through (:46-47).  ``mode`` is an addition: libwsu precision mode (see model/unet.py)."""
    if name.lower().startswith('unet'):
        nsteps = int(name.split('_')[1])
        return unet.UNet(
            in_channels=in_channels,
            out_channels=out_channels,
            nsteps=nsteps,
            drop_channel=channel,
            drop_rate=drop_rate,
            mode=mode,
        )
    raise NotImplementedError(name)


def load_model(
    model_path: pathlib.Path,
    model_name: str,
    device,
    *,
    network: str = 'unet_1',
    **kw
) -> torch.nn.Module:
    """Reference :52-62 hard-codes 'unet_1' regardless of the run's config; that stays the default here,
    ``network=`` lets callers pass the real depth.  Loads ``<model_path>/<model_name>/model/best_model.pt.tar``."""
    model = get_model(network, **kw).to(device)
    resume_model_file = pathlib.Path(model_path) / model_name / 'model' / 'best_model.pt.tar'
    checkpoint = torch.load(resume_model_file, map_location=device, weights_only=True)
    model.load_state_dict(checkpoint['state_dict'])
    return model
