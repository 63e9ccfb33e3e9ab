"""Helpers shared by the -m gpu parity tests: layout conversion and oracle plumbing."""
import numpy as np
import torch

from ws_unet_amd import formula, ops
from ws_unet_amd.model import get_model
from oracle import unet_ref

DEV = "cuda"
MODE_NAMES = ["f32", "bf16x3", "bf16"]
DEFAULT_MODE = "f16f4p"                               # what get_model(mode=None) / get_pretrained build (planar storage, block-scaled fp4 cross terms in the 3x3 convs)
NET_MODES = MODE_NAMES + ["f16f8", "f16f8p", "f16f4p"]          # whole-network tests also cover the default inference mode (its storage format is opaque)
# absolute tolerance on the [0,1] sigmoid output / relative tolerance on activations, per precision mode
OUT_ATOL = {"f32": 4e-6, "bf16x3": 2e-5, "bf16": 3e-2, "f16f8": 1e-4, "f16f8p": 1e-4, "f16f4p": 6e-4, "f16f8q": 4e-4}        # f16f4p: measured max 2.6e-4 (MAE 2.5e-5); f16f8q: plain-f16 activations into d*1: measured max ~2e-4, mean 4e-5        # f16f8: measured max 4.5e-5 (mean 4e-6)
# deeper nets (K up to 9*2048 terms) accumulate more fp32 summation-order noise vs oneDNN
OUT_ATOL_DEEP = {"f32": 1e-5, "bf16x3": 4e-5, "bf16": 5e-2, "f16f8": 1.5e-4, "f16f8p": 1.5e-4, "f16f4p": 9e-4, "f16f8q": 6e-4}
ACT_RTOL = {"f32": 2e-5, "bf16x3": 1e-4, "bf16": 6e-2}


def to_nhwc(x_nchw: torch.Tensor, mode: str) -> torch.Tensor:
    """CPU NCHW fp32 -> device NHWC in the activation dtype of ``mode``."""
    dt = ops.act_dtype(ops.mode_id(mode))
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(DEV).to(dt)


def from_nhwc(y: torch.Tensor) -> torch.Tensor:
    """device NHWC (any dtype) -> CPU NCHW fp32."""
    return y.float().permute(0, 3, 1, 2).contiguous().cpu()


def rand_act(shape, key, scale=1.0, relu=True):
    """Deterministic NCHW activation tensor (post-ReLU like real inputs of the 3x3 convs)."""
    a = formula.formula_tensor(key, shape, scale)
    if relu:
        a = np.maximum(a, 0)
    return torch.from_numpy(a)


def images01(n, h, w, seed):
    u8 = formula.synthetic_images(n, h, w, seed)
    return u8, torch.from_numpy(u8.astype(np.float32) / np.float32(255.))[:, None]


def gpu_model(nsteps, variant="he", mode="f32", drop_rate=None):
    m = get_model(f"unet_{nsteps}", in_channels=1, out_channels=1, channel=[0], drop_rate=drop_rate, mode=mode)
    sd = formula.formula_state_dict(nsteps, variant)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(DEV)


def oracle_forward(x, nsteps, variant="he", intermediates=None):
    sd = unet_ref.to_torch_state(formula.formula_state_dict(nsteps, variant))
    with torch.no_grad():
        return unet_ref.unet_forward(x.clone(), sd, nsteps, intermediates=intermediates)


def err_stats(got: torch.Tensor, ref: torch.Tensor):
    d = (got.double() - ref.double()).abs()
    return {"max": d.max().item(), "mean": d.mean().item(), "refmax": ref.abs().max().item()}


def unsplit(t_nhwc: torch.Tensor) -> torch.Tensor:
    """'bf16x3s' storage -> fp32 values hi + lo (test-side restatement of the layout in include/wsu.h: per pixel and 16-channel
    chunk  hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15, bf16 each)."""
    raw = t_nhwc.detach().cpu().contiguous().numpy().view(np.uint16)            # (N,H,W,2C) halves
    n, h, w, c2 = raw.shape
    c = c2 // 2
    q = raw.reshape(n, h, w, c // 16, 2, 2, 8)                                   # chunk, (hi|lo), half, channel-in-half
    f = (q.astype(np.uint32) << 16).view(np.float32)
    return torch.from_numpy((f[..., 0, :, :] + f[..., 1, :, :]).reshape(n, h, w, c).copy())


def _e4m3_table() -> np.ndarray:
    v = np.arange(256)
    e, m = (v >> 3) & 15, v & 7
    mag = np.where(e == 0, m * 2.0 ** -9, (1 + m / 8.0) * 2.0 ** (e.astype(np.float64) - 7))
    return np.where(v & 0x80, -mag, mag).astype(np.float32)


def unsplit_f16f8(t_nhwc: torch.Tensor) -> torch.Tensor:
    """'f16f8' storage -> f16 part + residual part as fp32 (test-side restatement of include/wsu.h: 3 bytes per element; per pixel and
    16-channel chunk  f16 ch 0-7 | f16 ch 8-15 | e4m3((x - f16 x) * 2^12) ch 0-15  = 48 bytes)."""
    raw = t_nhwc.detach().cpu().contiguous().numpy().view(np.uint8)             # (N,H,W,3C) bytes
    n, h, w, c3 = raw.shape
    c = c3 // 3
    q = raw.reshape(n, h, w, c // 16, 48)
    hi = q[..., :32].copy().view(np.float16).astype(np.float32)                 # (..., 16)
    lo = _e4m3_table()[q[..., 32:48]] * 2.0 ** -12
    return torch.from_numpy((hi + lo).reshape(n, h, w, c).copy())


def planar_encode(x_nchw: torch.Tensor, lo_scale: float = 4096.0) -> torch.Tensor:
    """fp32 NCHW (any device) -> planar 'F16F8P' storage on the device (test-side restatement of include/wsu.h: [n][C/16][3 planes][H][W]
    [16 B]; plane 0/1 = f16 of channels 0-7 / 8-15, plane 2 = e4m3((x - f16 x) * 2^12) of channels 0-15; gradients: lo_scale = 2^14)."""
    x = x_nchw.to(DEV).float()
    n, c, h, w = x.shape
    xc = x.reshape(n, c // 16, 16, h, w).permute(0, 1, 3, 4, 2).contiguous()              # (n, chunk, h, w, 16)
    hi = xc.to(torch.float16)
    lo = ((xc - hi.float()) * lo_scale).clamp(-448, 448).to(torch.float8_e4m3fn)
    hb = hi.view(torch.uint8).reshape(n, c // 16, h, w, 2, 16)                            # two 16-byte f16 granules
    planes = torch.stack([hb[..., 0, :], hb[..., 1, :], lo.view(torch.uint8)], dim=2)     # (n, chunk, 3, h, w, 16)
    return planes.contiguous().view(torch.float32)


GRAD_LO = 16384.0          # residual scaling of planar gradients


def planar_decode(t: torch.Tensor, lo_scale: float = 4096.0, f16_only: bool = False) -> torch.Tensor:
    """planar 'F16F8P' storage -> fp32 NCHW on the CPU (f16 part + residual part).  f16_only: a gradient tensor written with products 'f16'
    carries no residual plane (its plane 2 is uninitialised memory)."""
    raw = t.detach().contiguous().view(torch.uint8)                                        # (n, chunk, 3, h, w, 16)
    n, nch, _, h, w, _ = raw.shape
    hi = torch.stack([raw[:, :, 0], raw[:, :, 1]], dim=-2).contiguous().view(torch.float16).reshape(n, nch, h, w, 16).float()
    lo = 0.0 if f16_only else raw[:, :, 2].contiguous().view(torch.float8_e4m3fn).float() / lo_scale
    return (hi + lo).permute(0, 1, 4, 2, 3).reshape(n, nch * 16, h, w).cpu()


# ---- planar Q storage ('F16F4P', round 4): test-side restatement of include/wsu.h K1q ---------------------------------------------------------
_FP4_VALUES = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])


def fp4_codes(v: torch.Tensor) -> torch.Tensor:
    """fp32 -> e2m1 codes (uint8, bit 3 = sign, bits 0-2 = index into {0, .5, 1, 1.5, 2, 3, 4, 6}): round to nearest even, saturating
    (v_cvt_scalef32_pk_fp4_*, tools/fp4_probe.hip)."""
    a = v.abs().double()
    e = torch.floor(torch.log2(a.clamp_min(1e-30))).clamp(0, 2)
    step = torch.exp2(e - 1)
    q = (torch.round(a / step) * step).clamp_max(6.0)                           # torch.round: half to even = even mantissa on this grid
    idx = torch.bucketize(q.float(), _FP4_VALUES.to(q.device))                  # exact grid values -> their index
    return (idx.to(torch.uint8) | ((v < 0) | ((v == 0) & (torch.signbit(v)))).to(torch.uint8) * 8)


def fp4_values(codes: torch.Tensor) -> torch.Tensor:
    mag = _FP4_VALUES.to(codes.device)[(codes & 7).long()]
    return torch.where((codes & 8) != 0, -mag, mag)


def q_block_exp(amax: torch.Tensor) -> torch.Tensor:
    """E of a block whose largest |f16 part| is amax (wsu_q4_block_exp): exponent field - 16, one less when the largest / 2^E would lie in [2, 3]."""
    ef = torch.where(amax >= 2.0 ** -14, torch.floor(torch.log2(amax.clamp_min(1e-30))) + 15, torch.zeros_like(amax))
    finer = (ef > 0) & (amax <= 1.5 * torch.exp2(ef - 15))
    return ef - 16 - finer.to(ef.dtype)


def _q_sblock_index(h, w, device):
    """index of pixel (y, x)'s scale byte inside a chunk's scale plane: 16 x 32-pixel tile blocks"""
    yy, xx = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
    tx = (w + 31) // 32
    return ((yy // 16) * tx + xx // 32) * 512 + (yy % 16) * 32 + xx % 32


def planar_q_parts(x_nchw: torch.Tensor):
    """fp32 NCHW -> the pieces of its planar Q encoding, per (n, chunk, h, w, 16): f16 parts, hi codes, residual codes; per (n, chunk, h, w): E."""
    x = x_nchw.float()
    n, c, h, w = x.shape
    xc = x.reshape(n, c // 16, 16, h, w).permute(0, 1, 3, 4, 2).contiguous()
    hi = xc.to(torch.float16)
    res = xc - hi.float()
    e = q_block_exp(hi.float().abs().amax(dim=-1))
    sc = torch.exp2(e)[..., None]
    return hi, fp4_codes(hi.float() / sc), fp4_codes(res * 2048.0 / sc), e


def planar_q_encode(x_nchw: torch.Tensor):
    """fp32 NCHW (any device) -> ops.PlanarQ on the device, as a producing epilogue writes it."""
    x = x_nchw.to(DEV).float()
    n, c, h, w = x.shape
    hi, ch, cr, e = planar_q_parts(x)
    hb = hi.view(torch.uint8).reshape(n, c // 16, h, w, 2, 16)
    codes = torch.cat([ch, cr], dim=-1)                                         # 32 nibbles per (pixel, chunk): hi 0-15, residual 0-15
    q = (codes[..., 0::2] | (codes[..., 1::2] << 4)).to(torch.uint8)             # low nibble first
    out = ops.PlanarQ.empty(n, c, h, w, x.device)
    out.data.zero_()
    hw = h * w
    flat = out.data
    flat[:, :, 0:16 * hw] = hb[..., 0, :].reshape(n, c // 16, -1)
    flat[:, :, 16 * hw:32 * hw] = hb[..., 1, :].reshape(n, c // 16, -1)
    flat[:, :, 32 * hw:48 * hw] = q.reshape(n, c // 16, -1)
    sidx = (48 * hw + _q_sblock_index(h, w, x.device)).reshape(-1)
    flat[:, :, sidx] = (e + 127).to(torch.uint8).reshape(n, c // 16, -1)
    return out


def planar_q_decode(t, parts: bool = False):
    """ops.PlanarQ -> fp32 NCHW on the CPU: f16 part + fp4 residual * 2^(E - 11) (the residual carries ~2.5 bits: a Q tensor holds less than the
    e4m3-residual format -- exactly what the fp4 conv multiplies).  parts=True: also (f16 parts, hi codes, residual codes, E)."""
    n, c, h, w = t.n, t.c, t.h, t.w
    nch, hw = c // 16, h * w
    raw = t.data.detach()
    h0 = raw[:, :, 0:16 * hw].reshape(n, nch, h, w, 16)
    h1 = raw[:, :, 16 * hw:32 * hw].reshape(n, nch, h, w, 16)
    hi = torch.stack([h0, h1], dim=-2).contiguous().view(torch.float16).reshape(n, nch, h, w, 16)
    q = raw[:, :, 32 * hw:48 * hw].reshape(n, nch, h, w, 16)
    codes = torch.stack([q & 15, q >> 4], dim=-1).reshape(n, nch, h, w, 32)
    sidx = (48 * hw + _q_sblock_index(h, w, raw.device)).reshape(-1)
    e = raw[:, :, sidx].reshape(n, nch, h, w).float() - 127
    val = hi.float() + fp4_values(codes[..., 16:]) * torch.exp2(e - 11)[..., None]
    val = val.permute(0, 1, 4, 2, 3).reshape(n, c, h, w).cpu()
    return (val, hi, codes[..., :16], codes[..., 16:], e) if parts else val
