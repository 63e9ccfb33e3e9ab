"""Thin torch-tensor wrappers over the C ABI (include/wsu.h).

PyTorch is plumbing here: it owns device memory (caching allocator) and the HIP
stream; every computation is a libwsu kernel.  Activations are NHWC tensors of
shape (N, H, W, C), fp32 for modes 'f32' / 'bf16x3', bf16 for mode 'bf16'.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import MODES, MODE_BF16, MODE_BF16X3, MODE_BF16X3S, MODE_F16F8, MODE_F16F8X, MODE_F16F8P, MODE_F16F8Q, MODE_F16F4P, check


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """HIP-event timer around libwsu launches on the stream they are launched on (torch's current stream).
    bench.py enables it for the timed region to get the dominant kernel's average launch duration."""

    def __init__(self):
        self.records = []          # (kernel, meta, start_event, end_event)

    def launch(self, kernel: str, meta: dict, fn):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn()
        e.record()
        if _layer is not None:
            meta = dict(meta, layer=_layer)
        self.records.append((kernel, meta, s, e))
        return rc

    def per_layer(self):
        """layer tag (set_layer) -> {'kernel', 'launches', 'total_ms', 'avg_ms', 'flops', 'bytes'} with flops / bytes PER LAUNCH;
        launches without a tag are skipped (call after a device sync)."""
        out = {}
        for k, meta, s, e in self.records:
            name = meta.get("layer")
            if name is None:
                continue
            d = out.setdefault(name, {"kernel": k, "launches": 0, "total_ms": 0.0, "flops": meta.get("flops", 0.0), "bytes": meta.get("bytes", 0.0),
                                      **{key: meta[key] for key in ("bytes_2B", "tiles", "steps_per_tile") if key in meta}})
            d["launches"] += 1
            d["total_ms"] += s.elapsed_time(e)
        for d in out.values():
            d["avg_ms"] = d["total_ms"] / max(1, d["launches"])
        return out

    def summary(self):
        """kernel -> {'launches', 'total_ms', 'avg_ms', 'flops', 'bytes'} (call after a device sync)."""
        out = {}
        for k, meta, s, e in self.records:
            d = out.setdefault(k, {"launches": 0, "total_ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["total_ms"] += s.elapsed_time(e)
            d["flops"] += meta.get("flops", 0.0)
            d["bytes"] += meta.get("bytes", 0.0)
        for d in out.values():
            d["avg_ms"] = d["total_ms"] / max(1, d["launches"])
        return out


_timer: Optional[KernelTimer] = None
_layer: Optional[str] = None


def set_layer(name: Optional[str]) -> None:
    """Label the launches that follow (UNet.forward_features names its layers); only read while a KernelTimer is installed."""
    global _layer
    _layer = name


def set_timer(t: Optional[KernelTimer]) -> None:
    global _timer
    _timer = t


def _launch(kernel: str, meta: dict, fn) -> int:
    return fn() if _timer is None else _timer.launch(kernel, meta, fn)


def _ptr(t) -> Optional[int]:
    return None if t is None else t.data_ptr()


def act_dtype(mode: int) -> torch.dtype:
    """dtype of an activation tensor's allocation.  'bf16x3s' / 'f16f8' tensors are float32-typed buffers holding the split encodings of
    include/wsu.h (bf16 hi / lo halves, 4 bytes per element; f16 + one e4m3 residual, 3 bytes per element -> ``store_channels``):
    only libwsu kernels read them."""
    return torch.bfloat16 if mode == MODE_BF16 else torch.float32


def store_channels(c: int, mode: int) -> int:
    """Last dimension of the float32-shaped allocation holding ``c`` channels: 'f16f8' keeps 3 bytes per element."""
    return c * 3 // 4 if mode == MODE_F16F8 else c


def logical_channels(t: torch.Tensor, mode: int) -> int:
    return t.shape[3] * 4 // 3 if mode == MODE_F16F8 else t.shape[3]


def _esz(mode: int) -> int:
    return 2 if mode == MODE_BF16 else (3 if mode == MODE_F16F8 else 4)


def weight_mode(mode: int) -> int:
    """The packed weights of 'bf16x3s' are the 'bf16x3' ones; 'f16f8' has its own packing, shared by 'f16f8x' (the same arithmetic on
    fp32 tensors: the training forward)."""
    return MODE_BF16X3 if mode == MODE_BF16X3S else (MODE_F16F8 if mode in (MODE_F16F8X, MODE_F16F8P, MODE_F16F8Q, MODE_F16F4P) else mode)


def first_layer_weight_mode(mode: int) -> int:
    """Packing of the 3x3 weights handed to conv3x3_fused_first (the layer's own mode; kept as the one place that decides it)."""
    return weight_mode(mode)


def _dev_check(*ts: Optional[torch.Tensor]) -> None:
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.WsuError("libwsu kernels run on the GPU only: got a CPU tensor (no CPU fallback exists)")
        if not t.is_contiguous():
            raise _lib.WsuError("libwsu kernels need contiguous tensors")


def mode_id(mode) -> int:
    if isinstance(mode, str):
        if mode not in MODES:
            raise ValueError(f"unknown precision mode {mode!r}; choose from {sorted(MODES)}")
        return MODES[mode]
    return int(mode)


# ---- weight packing ---------------------------------------------------------------------------------

def pack_conv3x3(w: torch.Tensor, mode: int, dgrad: bool = False) -> torch.Tensor:
    """w: (Cout, Cin, 3, 3) fp32 OIHW on the device -> packed byte buffer."""
    lib = _lib.load()
    w = w.detach()
    _dev_check(w)
    assert w.dtype == torch.float32 and w.dim() == 4 and w.shape[2:] == (3, 3)
    cout, cin = w.shape[:2]
    mode = weight_mode(mode)
    out = torch.empty(lib.wsu_conv3x3_packed_bytes(cin, cout, mode), dtype=torch.uint8, device=w.device)
    fn = lib.wsu_conv3x3_pack_dgrad if dgrad else lib.wsu_conv3x3_pack
    check(fn(w.data_ptr(), out.data_ptr(), cin, cout, mode, _stream()), "wsu_conv3x3_pack")
    return out


def pack_conv3x3_f4(w: torch.Tensor) -> torch.Tensor:
    """w: (Cout, Cin, 3, 3) fp32 OIHW on the device -> the packed weights of conv3x3_q: f16 planes + block-scaled fp4 cross-term
    granules + scale bytes (wsu_conv3x3_pack_f4)."""
    lib = _lib.load()
    w = w.detach()
    _dev_check(w)
    assert w.dtype == torch.float32 and w.dim() == 4 and w.shape[2:] == (3, 3)
    cout, cin = w.shape[:2]
    nbytes = lib.wsu_conv3x3_packed_f4_bytes(cin, cout)
    if nbytes == 0:
        raise _lib.WsuError(f"fp4 packing needs cin % 16 == 0 and cout % 64 == 0 (got {cin}, {cout})")
    out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    check(lib.wsu_conv3x3_pack_f4(w.data_ptr(), out.data_ptr(), cin, cout, _stream()), "wsu_conv3x3_pack_f4")
    return out


def pack_convt2x2(w: torch.Tensor, mode: int) -> torch.Tensor:
    """w: (Cin, Cout, 2, 2) fp32 on the device -> packed byte buffer."""
    lib = _lib.load()
    w = w.detach()
    _dev_check(w)
    assert w.dtype == torch.float32 and w.dim() == 4 and w.shape[2:] == (2, 2)
    cin, cout = w.shape[:2]
    mode = weight_mode(mode)
    out = torch.empty(lib.wsu_convt2x2_packed_bytes(cin, cout, mode), dtype=torch.uint8, device=w.device)
    check(lib.wsu_convt2x2_pack(w.data_ptr(), out.data_ptr(), cin, cout, mode, _stream()), "wsu_convt2x2_pack")
    return out


# ---- forward ops --------------------------------------------------------------------------------------

def conv3x3(x1: torch.Tensor, x2: Optional[torch.Tensor], w_packed: torch.Tensor, bias: Optional[torch.Tensor],
            cout: int, mode: int, relu: bool = True, pool: bool = False, pool_idx: bool = False,
            pad_zero: bool = False):
    """Returns y, or (y, y_pool[, idx]) when ``pool``."""
    lib = _lib.load()
    _dev_check(x1, x2, w_packed, bias)
    n, h, w = x1.shape[:3]
    c1 = logical_channels(x1, mode)
    c2 = 0 if x2 is None else logical_channels(x2, mode)
    dt = act_dtype(mode)
    assert x1.dtype == dt and (x2 is None or (x2.dtype == dt and x2.shape[:3] == x1.shape[:3]))
    y = torch.empty((n, h, w, store_channels(cout, mode)), dtype=dt, device=x1.device)
    yp = idx = None
    if pool:
        yp = torch.empty((n, h // 2, w // 2, store_channels(cout, mode)), dtype=dt, device=x1.device)
        if pool_idx:
            idx = torch.empty((n, h // 2, w // 2, cout), dtype=torch.uint8, device=x1.device)
    esz = _esz(mode)
    meta = {"flops": 2.0 * 9 * (c1 + c2) * cout * n * h * w,
            "bytes": float(n * h * w * (c1 + c2 + cout) * esz + (n * h * w // 4 * cout * esz if pool else 0) + 9 * (c1 + c2) * cout * max(esz, 4 if mode == MODE_F16F8 else 0))}
    check(_launch("conv3x3", meta, lambda: lib.wsu_conv3x3_fwd(
        x1.data_ptr(), _ptr(x2), w_packed.data_ptr(), _ptr(bias), y.data_ptr(), _ptr(yp), _ptr(idx),
        n, h, w, c1, c2, cout, mode, int(relu), int(pad_zero), _stream())), "wsu_conv3x3_fwd")
    if not pool:
        return y
    return (y, yp, idx) if pool_idx else (y, yp)


def conv3x3_head(x1: torch.Tensor, x2: Optional[torch.Tensor], w_packed: torch.Tensor, bias: Optional[torch.Tensor],
                 head_w: torch.Tensor, head_b: Optional[torch.Tensor], mode: int, want_logit: bool = False, want_y: bool = False):
    """Last 3x3 conv (64 output channels) + ReLU fused with the 1x1 head + sigmoid.  Returns out (N,co,H,W) fp32
    [, logit][, y NHWC]."""
    lib = _lib.load()
    hw2 = head_w.detach().reshape(head_w.shape[0], -1)
    _dev_check(x1, x2, w_packed, bias, hw2, head_b)
    n, h, w = x1.shape[:3]
    c1 = logical_channels(x1, mode)
    c2 = 0 if x2 is None else logical_channels(x2, mode)
    cout, hc = hw2.shape[1], hw2.shape[0]
    out = torch.empty((n, hc, h, w), dtype=torch.float32, device=x1.device)
    logit = torch.empty_like(out) if want_logit else None
    y = torch.empty((n, h, w, store_channels(cout, mode)), dtype=act_dtype(mode), device=x1.device) if want_y else None
    esz = _esz(mode)
    meta = {"flops": 2.0 * 9 * (c1 + c2) * cout * n * h * w,
            "bytes": float(n * h * w * ((c1 + c2) * esz + hc * 4) + 9 * (c1 + c2) * cout * max(esz, 4 if mode == MODE_F16F8 else 0))}
    check(_launch("conv3x3", meta, lambda: lib.wsu_conv3x3_head_fwd(
        x1.data_ptr(), _ptr(x2), w_packed.data_ptr(), _ptr(bias), _ptr(y), hw2.data_ptr(), _ptr(head_b), out.data_ptr(), _ptr(logit),
        n, h, w, c1, c2, cout, hc, mode, _stream())), "wsu_conv3x3_head_fwd")
    res = [out]
    if want_logit:
        res.append(logit)
    if want_y:
        res.append(y)
    return res[0] if len(res) == 1 else tuple(res)


def conv3x3_fused_first(x_nchw: torch.Tensor, w1: torch.Tensor, b1: Optional[torch.Tensor], w_packed: torch.Tensor,
                        bias: Optional[torch.Tensor], cout: int, mode: int, relu: bool = True, pool: bool = False, pool_idx: bool = False):
    """e11 + e12 (+pool) in one launch (wsu_conv3x3_fused_first_fwd): x_nchw (N,1,H,W) fp32 -> y NHWC [, y_pool [, idx]]."""
    lib = _lib.load()
    w1 = w1.detach().contiguous()
    _dev_check(x_nchw, w1, b1, w_packed, bias)
    n, cin, h, w = x_nchw.shape
    assert cin == 1 and tuple(w1.shape) == (64, 1, 3, 3) and x_nchw.dtype == torch.float32
    y = torch.empty((n, h, w, store_channels(cout, mode)), dtype=act_dtype(mode), device=x_nchw.device)
    yp = torch.empty((n, h // 2, w // 2, store_channels(cout, mode)), dtype=act_dtype(mode), device=x_nchw.device) if pool else None
    idx = torch.empty((n, h // 2, w // 2, cout), dtype=torch.uint8, device=x_nchw.device) if (pool and pool_idx) else None
    esz = _esz(mode)
    meta = {"flops": 2.0 * 9 * 64 * cout * n * h * w,
            "bytes": float(n * h * w * (4 + cout * esz) + 9 * 64 * cout * max(esz, 4 if mode == MODE_F16F8 else 0) + (n * (h // 2) * (w // 2) * cout * esz if pool else 0))}
    check(_launch("conv3x3", meta, lambda: lib.wsu_conv3x3_fused_first_fwd(
        x_nchw.data_ptr(), w1.data_ptr(), _ptr(b1), w_packed.data_ptr(), _ptr(bias), y.data_ptr(), _ptr(yp), _ptr(idx),
        n, h, w, cout, mode, int(relu), _stream())), "wsu_conv3x3_fused_first_fwd")
    if pool:
        return (y, yp, idx) if pool_idx else (y, yp)
    return y


PLANAR_PLANES = 3            # stored planes per 16-channel chunk: f16 ch 0-7 | f16 ch 8-15 | e4m3 residuals ch 0-15


def planar_shape(n: int, c: int, h: int, w: int):
    """Allocation shape (float32-typed) of a planar 'F16F8P' activation tensor: [n][C/16][3 planes][H][W][16 B] (include/wsu.h)."""
    return (n, c // 16, PLANAR_PLANES, h, w, 4)


PLANAR_A, PLANAR_Q = 0, 1     # include/wsu.h WSU_PLANAR_*: the e4m3-residual planar format / the planar Q format of mode 'f16f4p'


class PlanarQ:
    """A planar Q activation tensor ('F16F4P' storage, include/wsu.h, round 4): per image and 16-channel chunk the planes f16 ch 0-7 | f16 ch
    8-15 | Q (32 fp4 nibbles per pixel: the f16 parts and the residuals * 2^11, both over the block's power-of-two scale) as [H][W][16 B], then
    one E8M0 scale byte per pixel in 16 x 32-pixel tile blocks.  `data`: uint8 (N, C/16, chunk_bytes).  Written by the producing kernel's
    epilogue (conv3x3_q / convt2x2_pl / conv3x3_first_pl with y_format=PLANAR_Q), read by conv3x3_q; opaque to everything else."""
    __slots__ = ("data", "n", "c", "h", "w")

    def __init__(self, data: torch.Tensor, n: int, c: int, h: int, w: int):
        self.data, self.n, self.c, self.h, self.w = data, n, c, h, w

    @staticmethod
    def chunk_bytes(h: int, w: int) -> int:
        return 48 * h * w + 512 * ((h + 15) // 16) * ((w + 31) // 32)

    @classmethod
    def empty(cls, n: int, c: int, h: int, w: int, device) -> "PlanarQ":
        assert c % 16 == 0
        return cls(torch.empty((n, c // 16, cls.chunk_bytes(h, w)), dtype=torch.uint8, device=device), n, c, h, w)

    @property
    def device(self):
        return self.data.device

    def data_ptr(self) -> int:
        return self.data.data_ptr()


def _planar_out(fmt: int, n: int, c: int, h: int, w: int, device):
    return PlanarQ.empty(n, c, h, w, device) if fmt == PLANAR_Q else torch.empty(planar_shape(n, c, h, w), dtype=torch.float32, device=device)


def conv3x3_q(x1: PlanarQ, x2: Optional[PlanarQ], w_packed_f4: torch.Tensor, bias: Optional[torch.Tensor], cout: int,
              relu: bool = True, pool: bool = False, want_y: bool = True,
              head_w: Optional[torch.Tensor] = None, head_b: Optional[torch.Tensor] = None, want_logit: bool = False,
              range_flag: Optional[torch.Tensor] = None, y_format: int = PLANAR_Q):
    """3x3 reflect conv (+ReLU, +2x2 max-pool, +1x1 head and sigmoid) in the fp4-cross-term arithmetic on planar Q activations
    (wsu_conv3x3_q_fwd, csrc/conv3x3_q.hip; the default inference mode 'f16f4p').  x1 / x2: PlanarQ; w_packed_f4 from pack_conv3x3_f4;
    y / y_pool: PlanarQ (y_format=PLANAR_Q) or e4m3-residual planar tensors (PLANAR_A: what convt2x2_pl reads; always for a y beside the head).
    Returns y [, y_pool] or, with head_w, out [, logit][, y]."""
    lib = _lib.load()
    hw2 = None if head_w is None else head_w.detach().reshape(head_w.shape[0], -1).contiguous()
    assert isinstance(x1, PlanarQ) and (x2 is None or isinstance(x2, PlanarQ)), "conv3x3_q reads planar Q tensors (ops.PlanarQ)"
    _dev_check(x1.data, None if x2 is None else x2.data, w_packed_f4, bias, hw2, head_b)
    n, h, w, c1 = x1.n, x1.h, x1.w, x1.c
    c2 = 0
    if x2 is not None:
        assert (x2.n, x2.h, x2.w) == (n, h, w)
        c2 = x2.c
    assert w_packed_f4.numel() * w_packed_f4.element_size() == int(lib.wsu_conv3x3_packed_f4_bytes(c1 + c2, cout)), "w_packed_f4 is not pack_conv3x3_f4 of (cin, cout)"
    hc = 0 if hw2 is None else hw2.shape[0]
    yf = PLANAR_A if hc else y_format
    y = _planar_out(yf, n, cout, h, w, x1.device) if want_y else None
    yp = _planar_out(yf, n, cout, h // 2, w // 2, x1.device) if pool else None
    out = torch.empty((n, hc, h, w), dtype=torch.float32, device=x1.device) if hc else None
    logit = torch.empty_like(out) if (hc and want_logit) else None
    act = n * h * w * ((c1 + c2) + (cout if want_y else 0)) + (n * (h // 2) * (w // 2) * cout if pool else 0)     # planar elements read + written
    meta = {"flops": 2.0 * 9 * (c1 + c2) * cout * n * h * w,
            "bytes": float(act * 49 / 16 + n * h * w * hc * 4 + 9 * (c1 + c2) * cout * 28 / 9),
            "bytes_2B": float(act * 2 + n * h * w * hc * 4 + 9 * (c1 + c2) * cout * 2),
            "tiles": n * ((h + 15) // 16) * ((w + 31) // 32) * (cout // 64), "steps_per_tile": (c1 + c2) // 16}
    check(_launch("conv3x3_q", meta, lambda: lib.wsu_conv3x3_q_fwd(
        x1.data_ptr(), _ptr(x2), w_packed_f4.data_ptr(), _ptr(bias), _ptr(y), _ptr(yp), _ptr(hw2), _ptr(head_b), _ptr(out), _ptr(logit), hc,
        n, h, w, c1, c2, cout, int(relu), yf, _ptr(range_flag), _stream())), "wsu_conv3x3_q_fwd")
    if hc:
        res = [out] + ([logit] if want_logit else []) + ([y] if want_y else [])
        return res[0] if len(res) == 1 else tuple(res)
    return (y, yp) if pool else y


def conv3x3_q_fused_first(x_nchw: torch.Tensor, w1: torch.Tensor, b1: Optional[torch.Tensor], w_packed_f4: torch.Tensor,
                          bias: Optional[torch.Tensor], cout: int, relu: bool = True, range_flag: Optional[torch.Tensor] = None):
    """e11 + e12 + pool of the default mode in one launch (wsu_conv3x3_q_fused_first_fwd): x_nchw (N,1,H,W) fp32 -> (y, y_pool) PlanarQ.  Bitwise
    conv3x3_first_pl(y_format=PLANAR_Q) followed by conv3x3_q(pool=True); xe11 never reaches HBM."""
    lib = _lib.load()
    w1 = w1.detach()
    _dev_check(x_nchw, w1, b1, w_packed_f4, bias)
    n, cin, h, w = x_nchw.shape
    assert cin == 1 and x_nchw.dtype == torch.float32 and x_nchw.is_contiguous()
    # w1: (64,1,3,3) OIHW, or already the tap-major table (9, 64) the kernel's scalar loads read (UNet caches it per weight version)
    assert tuple(w1.shape) in ((64, 1, 3, 3), (9, 64)) and w1.dtype == torch.float32
    w1t = w1.contiguous() if tuple(w1.shape) == (9, 64) else w1.reshape(64, 9).t().contiguous()
    b1 = torch.zeros(64, dtype=torch.float32, device=x_nchw.device) if b1 is None else b1.detach().contiguous()
    assert w_packed_f4.numel() == int(lib.wsu_conv3x3_packed_f4_bytes(64, cout)), "conv3x3_q_fused_first needs weights from pack_conv3x3_f4 of (64, cout)"
    y = PlanarQ.empty(n, cout, h, w, x_nchw.device)
    yp = PlanarQ.empty(n, cout, h // 2, w // 2, x_nchw.device)
    act = n * h * w * cout + n * (h // 2) * (w // 2) * cout
    meta = {"flops": 2.0 * 9 * 64 * cout * n * h * w + 2.0 * 9 * 64 * n * h * w,
            "bytes": float(n * h * w * 4 + act * 49 / 16 + 9 * 64 * cout * 28 / 9), "bytes_2B": float(n * h * w * 4 + act * 2 + 9 * 64 * cout * 2),
            "tiles": n * ((h + 15) // 16) * ((w + 31) // 32) * (cout // 64), "steps_per_tile": 4}
    check(_launch("conv3x3_q", meta, lambda: lib.wsu_conv3x3_q_fused_first_fwd(
        x_nchw.data_ptr(), w1t.data_ptr(), b1.data_ptr(), w_packed_f4.data_ptr(), _ptr(bias), y.data_ptr(), yp.data_ptr(),
        n, h, w, cout, int(relu), _ptr(range_flag), _stream())), "wsu_conv3x3_q_fused_first_fwd")
    return y, yp


def pack_conv3x3_up(w3: torch.Tensor, wt: torch.Tensor, bt: Optional[torch.Tensor], b3: Optional[torch.Tensor], want_dense: bool = False):
    """Weights of the fused decoder-block entry conv3x3_up_q (wsu_conv3x3_up_pack, csrc/conv3x3_qu.hip).  w3: (Cout, Cup + C2, 3, 3) fp32, the
    block's first conv (input channels [0, Cup) = the transposed conv's output, torch.cat([xu, skip]) order, unet.py:172,178,184); wt: (Cl, Cup, 2, 2)
    fp32, the nn.ConvTranspose2d weights; bt / b3: their biases.  Returns (w_skip_packed, w_low_packed, bias[, wc_dense]): the skip half packed
    like any conv3x3_q weight, the upsampled half as parity-class 2x2-tap weights combined in fp32 on the device, the combined bias, and on
    request the combined weights (Cout, Cl, 2, 2, 2, 2) [py][px][dy][dx]."""
    lib = _lib.load()
    w3, wt = w3.detach().contiguous(), wt.detach().contiguous()
    bt = None if bt is None else bt.detach().contiguous()
    b3 = None if b3 is None else b3.detach().contiguous()
    _dev_check(w3, wt, bt, b3)
    assert w3.dtype == torch.float32 and w3.dim() == 4 and w3.shape[2:] == (3, 3) and wt.dtype == torch.float32 and wt.dim() == 4 and wt.shape[2:] == (2, 2)
    cout, ctot = w3.shape[:2]
    cl, cup = wt.shape[:2]
    c2 = ctot - cup
    nbytes = lib.wsu_conv3x3_up_packed_bytes(cl, cout)
    if nbytes == 0 or c2 <= 0 or c2 % 16:
        raise _lib.WsuError(f"fused upsample packing needs cl % 16 == 0, c2 % 16 == 0 (> 0) and cout % 64 == 0 (got cl={cl}, cup={cup}, c2={c2}, cout={cout})")
    w_skip = pack_conv3x3_f4(w3[:, cup:].contiguous())
    w_low = torch.empty(nbytes, dtype=torch.uint8, device=w3.device)
    bias = torch.empty(cout, dtype=torch.float32, device=w3.device)
    dense = torch.empty((cout, cl, 2, 2, 2, 2), dtype=torch.float32, device=w3.device) if want_dense else None
    check(lib.wsu_conv3x3_up_pack(w3.data_ptr(), wt.data_ptr(), _ptr(bt), _ptr(b3), w_low.data_ptr(), bias.data_ptr(), _ptr(dense),
                                  cl, cup, c2, cout, _stream()), "wsu_conv3x3_up_pack")
    return (w_skip, w_low, bias, dense) if want_dense else (w_skip, w_low, bias)


def conv3x3_up_q(x_low: PlanarQ, x_skip: PlanarQ, w_skip_packed: torch.Tensor, w_low_packed: torch.Tensor, bias: torch.Tensor, cout: int,
                 relu: bool = True, range_flag: Optional[torch.Tensor] = None) -> PlanarQ:
    """relu(conv3x3_reflect(cat[conv_transpose2x2_s2(x_low), x_skip])) of a decoder block (unet.py:171-173, 177-179, 183-185) in one launch, in the
    arithmetic of the default inference mode (wsu_conv3x3_up_q_fwd, csrc/conv3x3_qu.hip): the upsampled half runs as a 2x2-tap conv on x_low with
    weights combined per output-pixel parity class -- the upsampled tensor never exists.  x_low: PlanarQ at (h/2, w/2); x_skip: PlanarQ at (h, w);
    weights from pack_conv3x3_up.  Returns y: PlanarQ."""
    lib = _lib.load()
    assert isinstance(x_low, PlanarQ) and isinstance(x_skip, PlanarQ), "conv3x3_up_q reads planar Q tensors (ops.PlanarQ)"
    _dev_check(x_low.data, x_skip.data, w_skip_packed, w_low_packed, bias)
    n, h, w, c2, cl = x_skip.n, x_skip.h, x_skip.w, x_skip.c, x_low.c
    assert (x_low.n, 2 * x_low.h, 2 * x_low.w) == (n, h, w), "x_low must have half the skip tensor's height and width"
    assert w_skip_packed.numel() == int(lib.wsu_conv3x3_packed_f4_bytes(c2, cout)), "w_skip_packed is not pack_conv3x3_f4 of (c2, cout)"
    assert w_low_packed.numel() == int(lib.wsu_conv3x3_up_packed_bytes(cl, cout)), "w_low_packed is not pack_conv3x3_up of (cl, cout)"
    assert bias is not None and bias.numel() == cout
    y = PlanarQ.empty(n, cout, h, w, x_skip.device)
    cup = cl // 2
    act = n * h * w * (c2 + cout) + n * (h // 2) * (w // 2) * cl
    # algorithmic work = the two reference ops it replaces: ConvTranspose2d (2 * 4 * cl * cup MACs per low pixel) + the 3x3 conv over cup + c2 channels
    meta = {"flops": 2.0 * 9 * (cup + c2) * cout * n * h * w + 2.0 * 4 * cl * cup * n * (h // 2) * (w // 2),
            "flops_executed": 2.0 * (9 * c2 + 4 * cl) * cout * n * h * w,
            "bytes": float(act * 49 / 16 + (9 * c2 + 16 * cl) * cout * 28 / 9), "bytes_2B": float(act * 2 + (9 * (cup + c2) * cout + 4 * cl * cup) * 2),
            "tiles": n * ((h + 15) // 16) * ((w + 31) // 32) * (cout // 64), "steps_per_tile": c2 // 16 + 2 * (cl // 16)}
    check(_launch("conv3x3_up_q", meta, lambda: lib.wsu_conv3x3_up_q_fwd(
        x_low.data_ptr(), x_skip.data_ptr(), w_skip_packed.data_ptr(), w_low_packed.data_ptr(), bias.data_ptr(), y.data_ptr(),
        n, h, w, cl, c2, cout, int(relu), _ptr(range_flag), _stream())), "wsu_conv3x3_up_q_fwd")
    return y


def conv3x3_pl(x1: torch.Tensor, x2: Optional[torch.Tensor], w_packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int,
               relu: bool = True, pool: bool = False, want_y: bool = True,
               head_w: Optional[torch.Tensor] = None, head_b: Optional[torch.Tensor] = None, want_logit: bool = False,
               range_flag: Optional[torch.Tensor] = None, x_residual=True, want_mask: bool = False):
    """3x3 reflect conv (+ReLU, +2x2 max-pool, +1x1 head and sigmoid) on planar F16F8P activations (wsu_conv3x3_pl_fwd).
    x1 / x2: planar tensors (N, C/16, 4, H, W, 4); w_packed from pack_conv3x3(mode f16f8) (the block-scaled fp4 cross terms of the default
    inference mode: conv3x3_q on PlanarQ tensors).  Returns y [, y_pool] or, with head_w, out [, logit][, y]."""
    lib = _lib.load()
    hw2 = None if head_w is None else head_w.detach().reshape(head_w.shape[0], -1).contiguous()
    _dev_check(x1, x2, w_packed, bias, hw2, head_b)
    assert x1.dtype == torch.float32 and x1.dim() == 6 and x1.shape[2] == PLANAR_PLANES and x1.shape[5] == 4 and x1.is_contiguous()
    n, nch1, _, h, w, _ = x1.shape
    c1 = nch1 * 16
    c2 = 0
    if x2 is not None:
        assert x2.dtype == torch.float32 and x2.dim() == 6 and x2.shape[0] == n and x2.shape[3:5] == x1.shape[3:5] and x2.is_contiguous()
        c2 = x2.shape[1] * 16
    y = torch.empty(planar_shape(n, cout, h, w), dtype=torch.float32, device=x1.device) if want_y else None
    yp = torch.empty(planar_shape(n, cout, h // 2, w // 2), dtype=torch.float32, device=x1.device) if pool else None
    hc = 0 if hw2 is None else hw2.shape[0]
    out = torch.empty((n, hc, h, w), dtype=torch.float32, device=x1.device) if hc else None
    logit = torch.empty_like(out) if (hc and want_logit) else None
    mask = relu_mask_alloc(n, cout, h, w, x1.device) if want_mask else None      # 1-bit ReLU mask of y for the next conv's data gradient (training)
    act = n * h * w * ((c1 + c2) + (cout if want_y else 0)) + (n * (h // 2) * (w // 2) * cout if pool else 0)     # planar elements read + written
    meta = {"flops": 2.0 * 9 * (c1 + c2) * cout * n * h * w,
            "bytes": float(act * 3 + n * h * w * hc * 4 + 9 * (c1 + c2) * cout * 4),
            # SURVEY 8d counts 2 bytes per activation element (bf16) and per weight: reported beside the format's own 3 B (bench.py per_layer)
            "bytes_2B": float(act * 2 + n * h * w * hc * 4 + 9 * (c1 + c2) * cout * 2),
            "tiles": n * ((h + 15) // 16) * ((w + 31) // 32) * (cout // 64), "steps_per_tile": (c1 + c2) // 16}
    check(_launch("conv3x3_pl", meta, lambda: lib.wsu_conv3x3_pl_fwd(
        x1.data_ptr(), _ptr(x2), w_packed.data_ptr(), _ptr(bias), _ptr(y), _ptr(yp), _ptr(hw2), _ptr(head_b), _ptr(out), _ptr(logit), hc,
        n, h, w, c1, c2, cout, int(relu), int(x_residual), _ptr(range_flag), _ptr(mask), _stream())), "wsu_conv3x3_pl_fwd")
    if hc:
        res = [out] + ([logit] if want_logit else []) + ([y] if want_y else [])
        return res[0] if len(res) == 1 else tuple(res)
    if want_mask:
        return y, mask
    return (y, yp) if pool else y


def conv3x3_pl_fused_first(x_nchw: torch.Tensor, w1: torch.Tensor, b1: Optional[torch.Tensor], w_packed: torch.Tensor,
                           bias: Optional[torch.Tensor], cout: int, relu: bool = True, pool: bool = False,
                           range_flag: Optional[torch.Tensor] = None):
    """e11 + e12 (+pool) of the planar path in one launch (wsu_conv3x3_pl_fused_first_fwd): x_nchw (N,1,H,W) fp32 -> y planar [, y_pool]."""
    lib = _lib.load()
    w1 = w1.detach().contiguous()
    _dev_check(x_nchw, w1, b1, w_packed, bias)
    n, cin, h, w = x_nchw.shape
    assert cin == 1 and tuple(w1.shape) == (64, 1, 3, 3) and x_nchw.dtype == torch.float32 and x_nchw.is_contiguous()
    # the fused kernel multiplies e4m3 cross terms: it reads the 36 KB (block, chunk) slices of pack_conv3x3(mode f16f8), not the fp4 packing
    assert w_packed.numel() * w_packed.element_size() == int(lib.wsu_conv3x3_packed_bytes(64, cout, _lib.MODE_F16F8)), \
        "conv3x3_pl_fused_first needs weights from pack_conv3x3(w, mode_id('f16f8'))"
    y = torch.empty(planar_shape(n, cout, h, w), dtype=torch.float32, device=x_nchw.device)
    yp = torch.empty(planar_shape(n, cout, h // 2, w // 2), dtype=torch.float32, device=x_nchw.device) if pool else None
    act = n * h * w * cout + (n * (h // 2) * (w // 2) * cout if pool else 0)
    meta = {"flops": 2.0 * 9 * 64 * cout * n * h * w,
            "bytes": float(n * h * w * 4 + act * 3 + 9 * 64 * cout * 4), "bytes_2B": float(n * h * w * 4 + act * 2 + 9 * 64 * cout * 2),
            "tiles": n * ((h + 15) // 16) * ((w + 31) // 32) * (cout // 64), "steps_per_tile": 4}
    check(_launch("conv3x3_pl", meta, lambda: lib.wsu_conv3x3_pl_fused_first_fwd(
        x_nchw.data_ptr(), w1.data_ptr(), _ptr(b1), w_packed.data_ptr(), _ptr(bias), y.data_ptr(), _ptr(yp),
        n, h, w, cout, int(relu), _ptr(range_flag), _stream())), "wsu_conv3x3_pl_fused_first_fwd")
    return (y, yp) if pool else y


def convt2x2_pl(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int,
                range_flag: Optional[torch.Tensor] = None, y_format: int = PLANAR_A):
    """nn.ConvTranspose2d(k=2, s=2) + bias on planar F16F8P activations (wsu_convt2x2_pl_fwd); w_packed from pack_convt2x2(mode f16f8).
    y_format=PLANAR_Q: the result is a PlanarQ (the input of conv3x3_q)."""
    lib = _lib.load()
    _dev_check(x, w_packed, bias)
    assert x.dtype == torch.float32 and x.dim() == 6 and x.shape[2] == PLANAR_PLANES and x.shape[5] == 4 and x.is_contiguous()
    n, nch, _, h, w, _ = x.shape
    cin = nch * 16
    y = _planar_out(y_format, n, cout, 2 * h, 2 * w, x.device)
    meta = {"flops": 2.0 * 4 * cin * cout * n * h * w, "bytes": float(n * h * w * (cin + 4 * cout) * 3 + 4 * cin * cout * 4),
            "bytes_2B": float(n * h * w * (cin + 4 * cout) * 2 + 4 * cin * cout * 2),
            "tiles": n * ((h + 3) // 4) * ((w + 31) // 32) * (cout // 64), "steps_per_tile": cin // 32}
    check(_launch("convt2x2_pl", meta, lambda: lib.wsu_convt2x2_pl_fwd(
        x.data_ptr(), w_packed.data_ptr(), _ptr(bias), y.data_ptr(), n, h, w, cin, cout, y_format, _ptr(range_flag), _stream())), "wsu_convt2x2_pl_fwd")
    return y


def relu_mask_alloc(n: int, c: int, h: int, w: int, device) -> torch.Tensor:
    """A relu_mask plane (include/wsu.h): uint8 (N, C/8, 16 * ceil(h / 16), 32 * ceil(w / 32)); rows / columns beyond the image are padding
    that nothing reads as a mask of a stored pixel -- zeroed so that the allocation is deterministic."""
    return torch.zeros((n, c // 8, (h + 15) // 16 * 16, (w + 31) // 32 * 32), dtype=torch.uint8, device=device)


def conv3x3_first_pl(x_nchw: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], relu: bool = True,
                     range_flag: Optional[torch.Tensor] = None, want_mask: bool = False, y_format: int = PLANAR_A):
    """First layer (N, cin <= 8, H, W) fp32 -> planar F16F8P tensor with w.shape[0] channels (wsu_conv3x3_first_pl_fwd) [, its relu_mask plane];
    y_format=PLANAR_Q: a PlanarQ."""
    lib = _lib.load()
    w = w.detach().contiguous()
    _dev_check(x_nchw, w, bias)
    n, cin, h, wd = x_nchw.shape
    cout = w.shape[0]
    assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous() and w.shape[1] == cin
    y = _planar_out(y_format, n, cout, h, wd, x_nchw.device)
    mask = relu_mask_alloc(n, cout, h, wd, x_nchw.device) if want_mask else None
    meta = {"flops": 2.0 * 9 * cin * cout * n * h * wd, "bytes": float(n * h * wd * (cin * 4 + cout * 3)), "bytes_2B": float(n * h * wd * (cin * 4 + cout * 2))}
    check(_launch("conv3x3_first_pl", meta, lambda: lib.wsu_conv3x3_first_pl_fwd(
        x_nchw.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), n, h, wd, cin, cout, int(relu), y_format, _ptr(range_flag), _ptr(mask), _stream())), "wsu_conv3x3_first_pl_fwd")
    return (y, mask) if want_mask else y


def pack_conv3x3_ring(w: torch.Tensor) -> torch.Tensor:
    """The four border-ring weight sets of the planar data gradient (wsu_conv3x3_pack_ring)."""
    lib = _lib.load()
    w = w.detach()
    _dev_check(w)
    cout, cin = w.shape[:2]
    out = torch.empty(4 * lib.wsu_conv3x3_packed_bytes(cin, cout, MODE_F16F8), dtype=torch.uint8, device=w.device)
    check(lib.wsu_conv3x3_pack_ring(w.data_ptr(), out.data_ptr(), cin, cout, _stream()), "wsu_conv3x3_pack_ring")
    return out


PRODUCTS = {"f16f8": 0, "f16": 1}                     # wsu.h WSU_PRODUCTS_F16F8 / WSU_PRODUCTS_F16


def products_id(name: str) -> int:
    if name not in PRODUCTS:
        raise ValueError(f"products must be one of {sorted(PRODUCTS)} (got {name!r})")
    return PRODUCTS[name]


def conv3x3_pl_bwd_data(g: torch.Tensor, w_packed_dgrad: torch.Tensor, w_packed_ring: Optional[torch.Tensor], cin: int, csplit: int,
                        mask1: Optional[torch.Tensor] = None, mask2: Optional[torch.Tensor] = None, pad_zero: bool = False,
                        mask1_bits: Optional[torch.Tensor] = None, mask2_bits: Optional[torch.Tensor] = None, products: str = "f16f8"):
    """Data gradient of the 3x3 conv on planar tensors (wsu_conv3x3_pl_bwd_data).  g: planar gradient (N, Cout/16, 3, H, W, 4); returns
    dx1 (csplit channels) and dx2 (cin - csplit channels, or None), planar gradients.  products: 'f16f8' | 'f16' (wsu.h WSU_PRODUCTS_*)."""
    lib = _lib.load()
    _dev_check(g, w_packed_dgrad, w_packed_ring, mask1, mask2, mask1_bits, mask2_bits)
    assert g.dtype == torch.float32 and g.dim() == 6 and g.shape[2] == PLANAR_PLANES and g.is_contiguous()
    n, nch, _, h, w, _ = g.shape
    for mb, cm in ((mask1_bits, csplit), (mask2_bits, cin - csplit)):
        assert mb is None or (mb.dtype == torch.uint8 and mb.is_contiguous() and tuple(mb.shape) == (n, cm // 8, (h + 15) // 16 * 16, (w + 31) // 32 * 32)), "relu_mask plane shape"
    cout = nch * 16
    dx1 = torch.empty(planar_shape(n, csplit, h, w), dtype=torch.float32, device=g.device)
    dx2 = torch.empty(planar_shape(n, cin - csplit, h, w), dtype=torch.float32, device=g.device) if csplit < cin else None
    nbytes = 0 if pad_zero else lib.wsu_conv3x3_pl_bwd_data_workspace_bytes(n, h, w, cin, cout)
    ws = workspace(nbytes, g.device) if nbytes else None
    meta = {"flops": 2.0 * 9 * cin * cout * n * h * w, "bytes": float(n * h * w * (cin * 5 + cout * (2 if products == "f16" else 3)))}
    check(_launch("conv3x3_pl_bwd_data", meta, lambda: lib.wsu_conv3x3_pl_bwd_data(
        g.data_ptr(), w_packed_dgrad.data_ptr(), _ptr(w_packed_ring), _ptr(ws), 0 if ws is None else ws.numel() * 4,
        dx1.data_ptr(), _ptr(dx2), csplit, _ptr(mask1), _ptr(mask2), _ptr(mask1_bits), _ptr(mask2_bits), n, h, w, cin, cout, int(pad_zero),
        products_id(products), _stream())), "wsu_conv3x3_pl_bwd_data")
    return dx1, dx2


def conv3x3_pl_bwd_weight(g: torch.Tensor, x1: torch.Tensor, x2: Optional[torch.Tensor], want_bias: bool = True, products: str = "f16f8"):
    """Weight / bias gradient of the 3x3 conv on planar operands (wsu_conv3x3_pl_bwd_weight): g planar gradient, x1 / x2 planar inputs.
    products: 'f16f8' | 'f16' (wsu.h WSU_PRODUCTS_*)."""
    lib = _lib.load()
    _dev_check(g, x1, x2)
    n, nco, _, h, w, _ = g.shape
    cout, c1 = nco * 16, x1.shape[1] * 16
    c2 = 0 if x2 is None else x2.shape[1] * 16
    dw = torch.empty((cout, c1 + c2, 3, 3), dtype=torch.float32, device=g.device)
    db = torch.empty(cout, dtype=torch.float32, device=g.device) if want_bias else None
    ws = workspace(lib.wsu_wgrad_workspace_bytes(cout, c1 + c2, 9), g.device)
    meta = {"flops": 2.0 * 9 * (c1 + c2) * cout * n * h * w,
            "bytes": float(n * h * w * (2 if products == "f16" else 3) * (cout * ((c1 + c2) // 64) + (c1 + c2) * (cout // 64)))}
    check(_launch("conv3x3_pl_bwd_weight", meta, lambda: lib.wsu_conv3x3_pl_bwd_weight(
        g.data_ptr(), x1.data_ptr(), _ptr(x2), dw.data_ptr(), _ptr(db), ws.data_ptr(), ws.numel() * 4,
        n, h, w, c1, c2, cout, products_id(products), _stream())), "wsu_conv3x3_pl_bwd_weight")
    return dw, db


def convt2x2_pl_bwd_weight(x: torch.Tensor, dy: torch.Tensor, want_bias: bool = True, products: str = "f16f8"):
    """Weight / bias gradient of the transposed conv on planar operands: x planar input (N, Cin/16, 3, h, w, 4), dy planar gradient at (2h, 2w)."""
    lib = _lib.load()
    _dev_check(x, dy)
    n, nci, _, h, w, _ = x.shape
    cin, cout = nci * 16, dy.shape[1] * 16
    assert dy.shape[3] == 2 * h and dy.shape[4] == 2 * w
    dw = torch.empty((cin, cout, 2, 2), dtype=torch.float32, device=x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device) if want_bias else None
    ws = workspace(lib.wsu_wgrad_workspace_bytes(cin, cout, 4), x.device)
    meta = {"flops": 2.0 * 4 * cin * cout * n * h * w, "bytes": float(n * h * w * 3 * (cin * (cout // 64) + 4 * cout * (cin // 64)))}
    check(_launch("convt2x2_pl_bwd_weight", meta, lambda: lib.wsu_convt2x2_pl_bwd_weight(
        x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _ptr(db), ws.data_ptr(), ws.numel() * 4, n, h, w, cin, cout, products_id(products), _stream())), "wsu_convt2x2_pl_bwd_weight")
    return dw, db


def pack_convt2x2_pl_dgrad(w: torch.Tensor) -> torch.Tensor:
    """w: (Cin, Cout, 2, 2) -> data-gradient weights of the planar transposed conv (wsu_convt2x2_pl_pack_dgrad)."""
    lib = _lib.load()
    w = w.detach()
    _dev_check(w)
    cin, cout = w.shape[:2]
    out = torch.empty(cin * cout * 16, dtype=torch.uint8, device=w.device)
    check(lib.wsu_convt2x2_pl_pack_dgrad(w.data_ptr(), out.data_ptr(), cin, cout, _stream()), "wsu_convt2x2_pl_pack_dgrad")
    return out


def convt2x2_pl_bwd_data(dy: torch.Tensor, w_packed_dgrad: torch.Tensor, cin: int, mask: Optional[torch.Tensor], products: str = "f16f8") -> torch.Tensor:
    """dy: planar gradient (N, Cout/16, 3, 2h, 2w, 4) -> dx planar gradient with cin channels at (h, w), masked by (mask > 0)."""
    lib = _lib.load()
    _dev_check(dy, w_packed_dgrad, mask)
    n, nco, _, oh, ow, _ = dy.shape
    h, w, cout = oh // 2, ow // 2, nco * 16
    dx = torch.empty(planar_shape(n, cin, h, w), dtype=torch.float32, device=dy.device)
    meta = {"flops": 2.0 * 4 * cin * cout * n * h * w, "bytes": float(n * h * w * (cout * 12 + cin * 5))}
    check(_launch("convt2x2_pl_bwd_data", meta, lambda: lib.wsu_convt2x2_pl_bwd_data(
        dy.data_ptr(), w_packed_dgrad.data_ptr(), dx.data_ptr(), _ptr(mask), n, h, w, cin, cout, products_id(products), _stream())), "wsu_convt2x2_pl_bwd_data")
    return dx


def maxpool2x2_pl_bwd(skip_g: Optional[torch.Tensor], dy_pool: torch.Tensor, act: torch.Tensor, products: str = "f16f8") -> torch.Tensor:
    """(skip_g + routed dy_pool) * (act > 0) on planar tensors; writes into skip_g when given.  products 'f16': the gradient tensors carry no
    residual plane (wsu.h, K7p notes)."""
    lib = _lib.load()
    _dev_check(skip_g, dy_pool, act)
    n, nch, _, h, w, _ = act.shape
    g = skip_g if skip_g is not None else torch.empty_like(act)
    gb = 2 if products == "f16" else 3                     # bytes per gradient element
    meta = {"bytes": float(n * nch * 16 * h * w * (3 + gb + gb / 4 + (gb if skip_g is not None else 0)))}
    check(_launch("maxpool2x2_pl_bwd", meta, lambda: lib.wsu_maxpool2x2_pl_bwd(
        _ptr(skip_g), dy_pool.data_ptr(), act.data_ptr(), g.data_ptr(), n, h, w, nch * 16, products_id(products), _stream())), "wsu_maxpool2x2_pl_bwd")
    return g


def conv1x1_sigmoid_pl_bwd(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, products: str = "f16f8"):
    """Head backward on a planar input: returns g (planar gradient), dw (cout, C, 1, 1), db (cout)."""
    lib = _lib.load()
    w2 = w.detach().reshape(w.shape[0], -1).contiguous()
    _dev_check(x, w2, out, dout)
    n, nch, _, h, wd, _ = x.shape
    c, cout = nch * 16, w2.shape[0]
    g = torch.empty_like(x)
    dw = torch.empty((cout, c, 1, 1), dtype=torch.float32, device=x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device)
    ws = workspace(lib.wsu_head_pl_bwd_workspace_bytes(c, cout), x.device)
    meta = {"bytes": float(n * h * wd * (c * (5 if products == "f16" else 6) + cout * 8))}
    check(_launch("conv1x1_sigmoid_pl_bwd", meta, lambda: lib.wsu_conv1x1_sigmoid_pl_bwd(
        x.data_ptr(), w2.data_ptr(), out.data_ptr(), dout.data_ptr(), g.data_ptr(), dw.data_ptr(), db.data_ptr(),
        ws.data_ptr(), ws.numel() * 4, n, h, wd, c, cout, products_id(products), _stream())), "wsu_conv1x1_sigmoid_pl_bwd")
    return g, dw, db


def colsum_pl(g: torch.Tensor, products: str = "f16f8") -> torch.Tensor:
    """Per-channel sums of a planar gradient."""
    lib = _lib.load()
    _dev_check(g)
    n, nch, _, h, w, _ = g.shape
    c = nch * 16
    db = torch.empty(c, dtype=torch.float32, device=g.device)
    ws = workspace(lib.wsu_chansum_pl_workspace_bytes(c), g.device)
    check(_launch("colsum_pl", {"bytes": float(n * c * h * w * 3)}, lambda: lib.wsu_colsum_pl(
        g.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel() * 4, n, h, w, c, products_id(products), _stream())), "wsu_colsum_pl")
    return db


def conv3x3_first_pl_bwd_weight(g: torch.Tensor, x_nchw: torch.Tensor, want_bias: bool = True, products: str = "f16f8"):
    """First-layer weight / bias gradient from a planar gradient; single input plane."""
    lib = _lib.load()
    _dev_check(g, x_nchw)
    n, nch, _, h, w, _ = g.shape
    c = nch * 16
    if x_nchw.shape[1] != 1:
        raise ValueError("the planar training path handles single-plane inputs (in_channels = 1); use train_mode 'f32' / 'bf16x3' otherwise")
    dw = torch.empty((c, 1, 3, 3), dtype=torch.float32, device=g.device)
    db = torch.empty(c, dtype=torch.float32, device=g.device) if want_bias else None
    ws = workspace(lib.wsu_chansum_pl_workspace_bytes(c), g.device)
    check(_launch("conv3x3_first_pl_bwd_weight", {"bytes": float(n * h * w * (c * (2 if products == "f16" else 3) + 4))}, lambda: lib.wsu_conv3x3_first_pl_bwd_weight(
        g.data_ptr(), x_nchw.data_ptr(), dw.data_ptr(), _ptr(db), ws.data_ptr(), ws.numel() * 4, n, h, w, c, products_id(products), _stream())),
        "wsu_conv3x3_first_pl_bwd_weight")
    return dw, db


def conv3x3_first_pl_bwd_data(g: torch.Tensor, w: torch.Tensor, products: str = "f16f8") -> torch.Tensor:
    """Input gradient of the first layer from a planar gradient (wsu_conv3x3_first_pl_bwd_data): g planar (N, C/16, 3, H, W, 4), w (C, cin, 3, 3)
    -> dx (N, cin, H, W) fp32 in g's scale."""
    lib = _lib.load()
    w = w.detach().contiguous()
    _dev_check(g, w)
    n, nch, _, h, wd, _ = g.shape
    c, cin = nch * 16, w.shape[1]
    assert tuple(w.shape) == (c, cin, 3, 3)
    dx = torch.empty((n, cin, h, wd), dtype=torch.float32, device=g.device)
    check(_launch("conv3x3_first_pl_bwd_data", {"bytes": float(n * h * wd * (c * (2 if products == "f16" else 3) + 4 * cin))}, lambda: lib.wsu_conv3x3_first_pl_bwd_data(
        g.data_ptr(), w.data_ptr(), dx.data_ptr(), n, h, wd, cin, c, products_id(products), _stream())), "wsu_conv3x3_first_pl_bwd_data")
    return dx


def pack_conv3x3_wino(w: torch.Tensor) -> torch.Tensor:
    """OIHW fp32 -> Winograd F(2,3) packed weights of wsu_conv3x3_wino_fwd (mode bf16x3)."""
    lib = _lib.load()
    w = w.detach().contiguous().float()
    _dev_check(w)
    cout, cin = w.shape[0], w.shape[1]
    nbytes = lib.wsu_conv3x3_wino_packed_bytes(cin, cout)
    if nbytes == 0:
        raise _lib.WsuError(f"Winograd packing needs cin % 16 == 0 and cout % 64 == 0 (got {cin}, {cout})")
    wp = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    check(lib.wsu_conv3x3_wino_pack(w.data_ptr(), wp.data_ptr(), cin, cout, _stream()), "wsu_conv3x3_wino_pack")
    return wp


def conv3x3_wino(x1: torch.Tensor, x2: Optional[torch.Tensor], w_packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int,
                 relu: bool = True, pool: bool = False, pool_idx: bool = False,
                 head_w: Optional[torch.Tensor] = None, head_b: Optional[torch.Tensor] = None, want_logit: bool = False, want_y: bool = True):
    """3x3 reflect conv (+ReLU, +pool, +head) in mode bf16x3 through the Winograd kernel.  Returns y [, y_pool [, idx]] or,
    with head_w, out [, logit][, y] like conv3x3_head."""
    lib = _lib.load()
    hw2 = None if head_w is None else head_w.detach().reshape(head_w.shape[0], -1)
    _dev_check(x1, x2, w_packed, bias, hw2, head_b)
    assert x1.dtype == torch.float32
    n, h, w, c1 = x1.shape
    c2 = 0 if x2 is None else x2.shape[3]
    y = torch.empty((n, h, w, cout), dtype=torch.float32, device=x1.device) if (want_y or hw2 is None) else None
    yp = torch.empty((n, h // 2, w // 2, cout), dtype=torch.float32, device=x1.device) if pool else None
    idx = torch.empty((n, h // 2, w // 2, cout), dtype=torch.uint8, device=x1.device) if (pool and pool_idx) else None
    hc = 0 if hw2 is None else hw2.shape[0]
    out = torch.empty((n, hc, h, w), dtype=torch.float32, device=x1.device) if hc else None
    logit = torch.empty_like(out) if (hc and want_logit) else None
    meta = {"flops": 2.0 * 9 * (c1 + c2) * cout * n * h * w,
            "bytes": float(n * h * w * ((c1 + c2) * 4 + (cout * 4 if y is not None else hc * 4)) + 9 * (c1 + c2) * cout * 4
                           + (n * (h // 2) * (w // 2) * cout * 4 if pool else 0))}
    check(_launch("conv3x3", meta, lambda: lib.wsu_conv3x3_wino_fwd(
        x1.data_ptr(), _ptr(x2), w_packed.data_ptr(), _ptr(bias), _ptr(y), _ptr(yp), _ptr(idx),
        _ptr(hw2), _ptr(head_b), _ptr(out), _ptr(logit), hc, n, h, w, c1, c2, cout, int(relu), _stream())), "wsu_conv3x3_wino_fwd")
    if hc:
        res = [out] + ([logit] if want_logit else []) + ([y] if want_y else [])
        return res[0] if len(res) == 1 else tuple(res)
    if pool:
        return (y, yp, idx) if pool_idx else (y, yp)
    return y


def conv3x3_first(x_nchw: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], mode: int, relu: bool = True) -> torch.Tensor:
    lib = _lib.load()
    _dev_check(x_nchw, w, bias)
    n, cin, h, wd = x_nchw.shape
    cout = w.shape[0]
    assert x_nchw.dtype == torch.float32 and w.dtype == torch.float32 and w.shape[1] == cin
    y = torch.empty((n, h, wd, cout), dtype=act_dtype(mode), device=x_nchw.device)
    esz = 2 if mode == MODE_BF16 else 4
    meta = {"flops": 2.0 * 9 * cin * cout * n * h * wd, "bytes": float(n * h * wd * (cin * 4 + cout * esz))}
    check(_launch("conv3x3_first", meta, lambda: lib.wsu_conv3x3_first_fwd(
        x_nchw.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), n, h, wd, cin, cout, mode, int(relu), _stream())),
        "wsu_conv3x3_first_fwd")
    return y


def maxpool2x2(x: torch.Tensor, mode: int, want_idx: bool = False):
    lib = _lib.load()
    _dev_check(x)
    n, h, w, c = x.shape
    y = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
    idx = torch.empty((n, h // 2, w // 2, c), dtype=torch.uint8, device=x.device) if want_idx else None
    check(lib.wsu_maxpool2x2_fwd(x.data_ptr(), y.data_ptr(), _ptr(idx), n, h, w, c, mode, _stream()), "wsu_maxpool2x2_fwd")
    return (y, idx) if want_idx else y


def convt2x2(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int, mode: int) -> torch.Tensor:
    lib = _lib.load()
    _dev_check(x, w_packed, bias)
    n, h, w = x.shape[:3]
    cin = logical_channels(x, mode)
    y = torch.empty((n, 2 * h, 2 * w, store_channels(cout, mode)), dtype=x.dtype, device=x.device)
    esz = _esz(mode)
    meta = {"flops": 2.0 * 4 * cin * cout * n * h * w, "bytes": float(n * h * w * (cin + 4 * cout) * esz + 4 * cin * cout * 4)}
    check(_launch("convt2x2", meta, lambda: lib.wsu_convt2x2_fwd(
        x.data_ptr(), w_packed.data_ptr(), _ptr(bias), y.data_ptr(), n, h, w, cin, cout, mode, _stream())), "wsu_convt2x2_fwd")
    return y


def conv1x1_sigmoid(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], mode: int, want_logit: bool = False):
    """x: (N,H,W,C) NHWC; w: (Cout, C[,1,1]) fp32; returns NCHW fp32 (N,Cout,H,W)."""
    lib = _lib.load()
    w2 = w.detach().reshape(w.shape[0], -1)
    _dev_check(x, w2, bias)
    n, h, wd, c = x.shape
    cout = w2.shape[0]
    out = torch.empty((n, cout, h, wd), dtype=torch.float32, device=x.device)
    logit = torch.empty_like(out) if want_logit else None
    esz = 2 if mode == MODE_BF16 else 4
    meta = {"flops": 2.0 * c * cout * n * h * wd, "bytes": float(n * h * wd * (c * esz + cout * 4))}
    check(_launch("conv1x1_sigmoid", meta, lambda: lib.wsu_conv1x1_sigmoid_fwd(
        x.data_ptr(), w2.data_ptr(), _ptr(bias), out.data_ptr(), _ptr(logit), n, h, wd, c, cout, mode, _stream())),
        "wsu_conv1x1_sigmoid_fwd")
    return (out, logit) if want_logit else out


def uniform_dropout(x: torch.Tensor, mask: Optional[torch.Tensor], channel: int = 0, keep_prob: float = 1.0,
                    seed: int = 0, want_mask: bool = False):
    """Out of place: returns y (and the mask used)."""
    lib = _lib.load()
    _dev_check(x, mask)
    n, c, h, w = x.shape
    assert x.dtype == torch.float32
    y = torch.empty_like(x)
    mo = torch.empty((n, 1, h, w), dtype=torch.float32, device=x.device) if want_mask else None
    check(lib.wsu_uniform_dropout_fwd(x.data_ptr(), y.data_ptr(), _ptr(mask), _ptr(mo), n, c, h, w, channel,
                                      float(keep_prob), int(seed) & 0xFFFFFFFFFFFFFFFF, _stream()), "wsu_uniform_dropout_fwd")
    return (y, mo) if want_mask else y


def ws_residual_stats(x_u8: torch.Tensor, y01: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """x_u8: (N,H,W) uint8; y01: (N,H,W) or (N,1,H,W) fp32 -> (beta_hat[N], l1[N]) fp32."""
    lib = _lib.load()
    _dev_check(x_u8, y01)
    n, h, w = x_u8.shape
    assert x_u8.dtype == torch.uint8 and y01.dtype == torch.float32 and y01.numel() == n * h * w
    beta = torch.empty(n, dtype=torch.float32, device=x_u8.device)
    l1 = torch.empty(n, dtype=torch.float32, device=x_u8.device)
    check(lib.wsu_ws_residual_stats(x_u8.data_ptr(), y01.data_ptr(), beta.data_ptr(), l1.data_ptr(), n, h, w, _stream()),
          "wsu_ws_residual_stats")
    return beta, l1


def _taps(k) -> Optional[np.ndarray]:
    """(3,3) / (3,3,1) kernel array -> 9 contiguous host floats K[a][b] (layout of the reference's NAMED_FILTERS)."""
    if k is None:
        return None
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 3 and k.shape[2] == 1:
        k = k[..., 0]
    if k.shape != (3, 3):
        raise ValueError(f"3x3 single-channel kernel expected, got shape {k.shape}")
    return np.ascontiguousarray(k)


def ws_attack(x_u8: torch.Tensor, x_hat: Optional[torch.Tensor] = None, *, x_bias: Optional[torch.Tensor] = None,
              pixel_filter=None, mean_filter=None, hat_scale: float = 255.0, weighted: int = 1, correct_bias: bool = False,
              return_sums: bool = False):
    """Batched WS payload estimate (wsu_ws_attack, src/ws/estimate.py:55-136).  x_u8: (N,H,W) uint8.
    x_hat: (N,H,W)/(N,1,H,W) full-frame prediction or (N,H-2,W-2) interior prediction, multiplied by `hat_scale`;
    or `pixel_filter` (3,3[,1]) for an in-kernel linear predictor.  Returns beta_hat[N] (and the (N,3) fp64 sums)."""
    lib = _lib.load()
    _dev_check(x_u8, *[t for t in (x_hat, x_bias) if t is not None])
    n, h, w = x_u8.shape
    assert x_u8.dtype == torch.uint8
    hat_full = 1
    if x_hat is not None:
        assert x_hat.dtype == torch.float32 and x_hat.is_contiguous()
        if x_hat.numel() == n * h * w:
            hat_full = 1
        elif x_hat.numel() == n * (h - 2) * (w - 2):
            hat_full = 0
        else:
            raise ValueError(f"prediction of {tuple(x_hat.shape)} does not match pixels {tuple(x_u8.shape)}")
        if x_bias is not None:
            assert x_bias.dtype == torch.float32 and x_bias.is_contiguous() and x_bias.numel() == x_hat.numel()
    pt, mt = _taps(pixel_filter), _taps(mean_filter)
    beta = torch.empty(n, dtype=torch.float32, device=x_u8.device)
    sums = torch.empty((n, 3), dtype=torch.float64, device=x_u8.device) if return_sums else None
    ws = torch.empty(lib.wsu_ws_attack_workspace_bytes(n) // 8, dtype=torch.float64, device=x_u8.device)
    check(_launch("ws_attack", {}, lambda: lib.wsu_ws_attack(
        x_u8.data_ptr(), x_hat.data_ptr() if x_hat is not None else None, x_bias.data_ptr() if x_bias is not None else None,
        pt.ctypes.data if pt is not None else None, mt.ctypes.data if mt is not None else None,
        hat_full, float(hat_scale), int(weighted), int(bool(correct_bias)), beta.data_ptr(),
        sums.data_ptr() if sums is not None else None, ws.data_ptr(), ws.numel() * 8, n, h, w, _stream())), "wsu_ws_attack")
    return (beta, sums) if return_sums else beta


def filter3x3_valid(x: torch.Tensor, kernel) -> torch.Tensor:
    """x: (N,H,W) fp32 -> (N,H-2,W-2) fp32 = convolve(x/255., K, 'valid')*255. (wsu_filter3x3_valid_f32)."""
    lib = _lib.load()
    _dev_check(x)
    assert x.dtype == torch.float32 and x.dim() == 3
    n, h, w = x.shape
    k = _taps(kernel)
    y = torch.empty((n, h - 2, w - 2), dtype=torch.float32, device=x.device)
    check(lib.wsu_filter3x3_valid_f32(x.data_ptr(), k.ctypes.data, y.data_ptr(), n, h, w, _stream()), "wsu_filter3x3_valid_f32")
    return y


def lsb_delta_unit(x_u8: torch.Tensor) -> torch.Tensor:
    """((x ^ 1) - x) / 255. as fp32, same shape."""
    lib = _lib.load()
    _dev_check(x_u8)
    y = torch.empty(x_u8.shape, dtype=torch.float32, device=x_u8.device)
    check(lib.wsu_lsb_delta_unit_f32(x_u8.data_ptr(), y.data_ptr(), x_u8.numel(), _stream()), "wsu_lsb_delta_unit_f32")
    return y


def ws_meter_beta(x01: torch.Tensor, y01: torch.Tensor) -> torch.Tensor:
    """WSMeter's per-image beta_hat (fp64, interior crop) from float inputs / outputs of shape (N,1,H,W) or (N,H,W)."""
    lib = _lib.load()
    _dev_check(x01, y01)
    assert x01.dtype == torch.float32 and y01.dtype == torch.float32 and x01.numel() == y01.numel()
    n, h, w = x01.shape[0], x01.shape[-2], x01.shape[-1]
    assert x01.numel() == n * h * w, "single-plane images expected"
    beta = torch.empty(n, dtype=torch.float64, device=x01.device)
    check(lib.wsu_ws_meter_beta(x01.data_ptr(), y01.data_ptr(), beta.data_ptr(), n, h, w, _stream()), "wsu_ws_meter_beta")
    return beta


def u8_to_unit(x_u8: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _dev_check(x_u8)
    y = torch.empty(x_u8.shape, dtype=torch.float32, device=x_u8.device)
    check(lib.wsu_u8_to_unit_f32(x_u8.data_ptr(), y.data_ptr(), x_u8.numel(), _stream()), "wsu_u8_to_unit_f32")
    return y


# ---- backward ops (fp32 storage; g = pre-activation gradient, NHWC) -------------------------------------

_ws_cache = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per device (fp32 view), reused by all backward reductions on the stream."""
    key = str(device)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


def pack_convt2x2_dgrad(w: torch.Tensor, mode: int) -> torch.Tensor:
    lib = _lib.load()
    w = w.detach()
    _dev_check(w)
    cin, cout = w.shape[:2]
    out = torch.empty(lib.wsu_convt2x2_packed_dgrad_bytes(cin, cout, mode), dtype=torch.uint8, device=w.device)
    check(lib.wsu_convt2x2_pack_dgrad(w.data_ptr(), out.data_ptr(), cin, cout, mode, _stream()), "wsu_convt2x2_pack_dgrad")
    return out


def conv3x3_bwd_data(g: torch.Tensor, w_packed_dgrad: torch.Tensor, w_oihw: torch.Tensor, csplit: int,
                     mask1: Optional[torch.Tensor], mask2: Optional[torch.Tensor], mode: int):
    """Returns dx1 (N,H,W,csplit) and dx2 (N,H,W,cin-csplit) or None."""
    lib = _lib.load()
    w_oihw = w_oihw.detach()
    _dev_check(g, w_packed_dgrad, w_oihw, mask1, mask2)
    n, h, w, cout = g.shape
    cin = w_oihw.shape[1]
    assert g.dtype == torch.float32 and w_oihw.shape[0] == cout
    dx1 = torch.empty((n, h, w, csplit), dtype=torch.float32, device=g.device)
    dx2 = torch.empty((n, h, w, cin - csplit), dtype=torch.float32, device=g.device) if csplit < cin else None
    meta = {"flops": 2.0 * 9 * cin * cout * n * h * w}
    nbytes = lib.wsu_conv3x3_bwd_data_workspace_bytes(n, h, w, cin, cout, mode)
    ws = workspace(nbytes, g.device)                          # border strips + tap-swapped weights (fp32 view, grow-only)
    check(_launch("conv3x3_bwd_data", meta, lambda: lib.wsu_conv3x3_bwd_data(
        g.data_ptr(), w_packed_dgrad.data_ptr(), w_oihw.data_ptr(), ws.data_ptr(), ws.numel() * 4, dx1.data_ptr(), _ptr(dx2), csplit,
        _ptr(mask1), _ptr(mask2), n, h, w, cin, cout, mode, _stream())), "wsu_conv3x3_bwd_data")
    return dx1, dx2


def conv3x3_bwd_weight(g: torch.Tensor, x1: torch.Tensor, x2: Optional[torch.Tensor], want_bias: bool = True, mode: int = 0):
    lib = _lib.load()
    _dev_check(g, x1, x2)
    n, h, w, cout = g.shape
    c1 = x1.shape[3]
    c2 = 0 if x2 is None else x2.shape[3]
    dw = torch.empty((cout, c1 + c2, 3, 3), dtype=torch.float32, device=g.device)
    db = torch.empty(cout, dtype=torch.float32, device=g.device) if want_bias else None
    nbytes = max(lib.wsu_wgrad_workspace_bytes(cout, c1 + c2, 9), (n * h * w + 4095) // 4096 * cout * 4)
    ws = workspace(nbytes, g.device)
    meta = {"flops": 2.0 * 9 * (c1 + c2) * cout * n * h * w}
    check(_launch("conv3x3_bwd_weight", meta, lambda: lib.wsu_conv3x3_bwd_weight(
        g.data_ptr(), x1.data_ptr(), _ptr(x2), dw.data_ptr(), _ptr(db), ws.data_ptr(), ws.numel() * 4,
        n, h, w, c1, c2, cout, mode, _stream())), "wsu_conv3x3_bwd_weight")
    return dw, db


def conv3x3_first_bwd_weight(g: torch.Tensor, x_nchw: torch.Tensor, want_bias: bool = True):
    lib = _lib.load()
    _dev_check(g, x_nchw)
    n, h, w, cout = g.shape
    cin = x_nchw.shape[1]
    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=g.device)
    db = torch.empty(cout, dtype=torch.float32, device=g.device) if want_bias else None
    ws = workspace(lib.wsu_first_bwd_workspace_bytes(n, h, w, cin, cout), g.device)
    check(_launch("conv3x3_first_bwd_weight", {}, lambda: lib.wsu_conv3x3_first_bwd_weight(
        g.data_ptr(), x_nchw.data_ptr(), dw.data_ptr(), _ptr(db), ws.data_ptr(), ws.numel() * 4, n, h, w, cin, cout, _stream())),
        "wsu_conv3x3_first_bwd_weight")
    return dw, db


def conv3x3_first_bwd_data(g: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """g: (N,H,W,cout) pre-activation gradient of the first layer -> dx (N,cin,H,W)."""
    lib = _lib.load()
    w = w.detach()
    _dev_check(g, w)
    n, h, wd, cout = g.shape
    cin = w.shape[1]
    dx = torch.empty((n, cin, h, wd), dtype=torch.float32, device=g.device)
    check(lib.wsu_conv3x3_first_bwd_data(g.data_ptr(), w.data_ptr(), dx.data_ptr(), n, h, wd, cin, cout, _stream()),
          "wsu_conv3x3_first_bwd_data")
    return dx


def convt2x2_bwd_weight(x: torch.Tensor, dy: torch.Tensor, want_bias: bool = True, mode: int = 0):
    lib = _lib.load()
    _dev_check(x, dy)
    n, h, w, cin = x.shape
    cout = dy.shape[3]
    assert dy.shape == (n, 2 * h, 2 * w, cout)
    dw = torch.empty((cin, cout, 2, 2), dtype=torch.float32, device=x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device) if want_bias else None
    nbytes = max(lib.wsu_wgrad_workspace_bytes(cin, cout, 4), (n * h * w * 4 + 4095) // 4096 * cout * 4)
    ws = workspace(nbytes, x.device)
    meta = {"flops": 2.0 * 4 * cin * cout * n * h * w}
    check(_launch("convt2x2_bwd_weight", meta, lambda: lib.wsu_convt2x2_bwd_weight(
        x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _ptr(db), ws.data_ptr(), ws.numel() * 4, n, h, w, cin, cout, mode, _stream())),
        "wsu_convt2x2_bwd_weight")
    return dw, db


def convt2x2_bwd_data(dy: torch.Tensor, w_packed_dgrad: torch.Tensor, cin: int, mask: Optional[torch.Tensor], mode: int) -> torch.Tensor:
    lib = _lib.load()
    _dev_check(dy, w_packed_dgrad, mask)
    n, oh, ow, cout = dy.shape
    h, w = oh // 2, ow // 2
    dx = torch.empty((n, h, w, cin), dtype=torch.float32, device=dy.device)
    meta = {"flops": 2.0 * 4 * cin * cout * n * h * w}
    check(_launch("convt2x2_bwd_data", meta, lambda: lib.wsu_convt2x2_bwd_data(
        dy.data_ptr(), w_packed_dgrad.data_ptr(), dx.data_ptr(), _ptr(mask), n, h, w, cin, cout, mode, _stream())),
        "wsu_convt2x2_bwd_data")
    return dx


def maxpool2x2_bwd(g_full: Optional[torch.Tensor], dy_pool: torch.Tensor, idx: torch.Tensor, xp_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """Adds the routed pooled gradient into ``g_full`` (allocated zero-initialised when None)."""
    lib = _lib.load()
    _dev_check(g_full, dy_pool, idx, xp_mask)
    n, hp, wp, c = dy_pool.shape
    accumulate = g_full is not None
    if g_full is None:
        g_full = torch.empty((n, 2 * hp, 2 * wp, c), dtype=torch.float32, device=dy_pool.device)
    h, w = g_full.shape[1:3]
    check(_launch("maxpool2x2_bwd", {}, lambda: lib.wsu_maxpool2x2_bwd(
        g_full.data_ptr(), dy_pool.data_ptr(), idx.data_ptr(), _ptr(xp_mask), n, h, w, c, int(accumulate), _stream())),
        "wsu_maxpool2x2_bwd")
    return g_full


def conv1x1_sigmoid_bwd(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, relu_mask: bool = True):
    """x: (N,H,W,C) saved input of the head; out/dout: (N,cout,H,W).  Returns gx (N,H,W,C), dw (cout,C,1,1), db (cout)."""
    lib = _lib.load()
    w2 = w.detach().reshape(w.shape[0], -1)
    _dev_check(x, w2, out, dout)
    n, h, wd, c = x.shape
    cout = w2.shape[0]
    gx = torch.empty_like(x)
    dw = torch.empty((cout, c, 1, 1), dtype=torch.float32, device=x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device)
    ws = workspace(lib.wsu_head_bwd_workspace_bytes(c, cout), x.device)
    check(_launch("conv1x1_sigmoid_bwd", {}, lambda: lib.wsu_conv1x1_sigmoid_bwd(
        x.data_ptr(), w2.data_ptr(), out.data_ptr(), dout.data_ptr(), gx.data_ptr(), dw.data_ptr(), db.data_ptr(),
        ws.data_ptr(), ws.numel() * 4, n, h, wd, c, cout, int(relu_mask), _stream())), "wsu_conv1x1_sigmoid_bwd")
    return gx, dw, db


def l1ws_loss_fwd_bwd(out: torch.Tensor, covers: torch.Tensor, inputs: torch.Tensor, alphas: torch.Tensor,
                      use_l1: int = 1, use_ws: bool = True):
    """Returns (loss scalar tensor, dLoss/dout, parts[l1, ws], beta_hat[N])."""
    lib = _lib.load()
    _dev_check(out, covers, inputs, alphas)
    assert out.shape == covers.shape == inputs.shape and out.dtype == torch.float32
    n = out.shape[0]
    per = out.numel() // n
    loss = torch.empty((), dtype=torch.float32, device=out.device)
    parts = torch.empty(2, dtype=torch.float32, device=out.device)
    beta = torch.empty(n, dtype=torch.float32, device=out.device)
    dout = torch.empty_like(out)
    ws = torch.empty((lib.wsu_l1ws_loss_workspace_bytes(n) + 7) // 8, dtype=torch.float64, device=out.device)
    alphas = alphas.to(torch.float32).contiguous()
    check(lib.wsu_l1ws_loss_fwd_bwd(out.data_ptr(), covers.data_ptr(), inputs.data_ptr(), alphas.data_ptr(), loss.data_ptr(),
                                    parts.data_ptr(), dout.data_ptr(), beta.data_ptr(), ws.data_ptr(), ws.numel() * 8,
                                    n, per, int(use_l1), int(use_ws), _stream()), "wsu_l1ws_loss_fwd_bwd")
    return loss, dout, parts, beta


class AdamWTable:
    """Device-side descriptor table for wsu_adamw_multi_tensor over a fixed list of (param, grad, m, v)."""

    def __init__(self, params, grads, exp_avg, exp_avg_sq):
        import numpy as np
        rows, first = [], 0
        for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
            _dev_check(p, g, m, v)
            assert p.dtype == g.dtype == m.dtype == v.dtype == torch.float32 and p.numel() == g.numel()
            rows.append([p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), first])
            first += (p.numel() + 1023) // 1024
        self.total_blocks = first
        self.ntensors = len(rows)
        self.table = torch.from_numpy(np.array(rows, dtype=np.int64)).to(params[0].device)
        self.ptrs = [r[:4] for r in rows]

    def matches(self, params, grads):
        return all(p.data_ptr() == r[0] and g.data_ptr() == r[1] for p, g, r in zip(params, grads, self.ptrs))

    def step(self, step: int, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0,
             skip_flag: Optional[torch.Tensor] = None):
        lib = _lib.load()
        check(lib.wsu_adamw_multi_tensor(self.table.data_ptr(), self.ntensors, self.total_blocks, lr, betas[0], betas[1],
                                         eps, weight_decay, step, grad_scale, _ptr(skip_flag), _stream()), "wsu_adamw_multi_tensor")


def pow2_grad_scale(x: torch.Tensor) -> torch.Tensor:
    """Device-side {scale, 1/scale} with scale = 2^floor(2 - log2 max|x|) (wsu_pow2_grad_scale): no host sync, no ATen arithmetic."""
    lib = _lib.load()
    _dev_check(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    ws = workspace(16, x.device)
    check(lib.wsu_pow2_grad_scale(x.data_ptr(), x.numel(), out.data_ptr(), ws.data_ptr(), _stream()), "wsu_pow2_grad_scale")
    return out


def scale_by(x: torch.Tensor, factor: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x * factor[0] with the factor on the device (wsu_scale_f32); ``out`` may be ``x`` itself."""
    lib = _lib.load()
    _dev_check(x, factor)
    assert x.dtype == torch.float32 and x.is_contiguous() and factor.dtype == torch.float32
    y = torch.empty_like(x) if out is None else out
    check(lib.wsu_scale_f32(x.data_ptr(), y.data_ptr(), x.numel(), factor.data_ptr(), _stream()), "wsu_scale_f32")
    return y


def scale_many_(tensors, factor: torch.Tensor) -> None:
    """In-place multiplication of every tensor by factor[0] in ONE launch (wsu_scale_multi_tensor)."""
    import numpy as np
    lib = _lib.load()
    tensors = [t for t in tensors if t is not None]
    if not tensors:
        return
    rows, first = [], 0
    for t in tensors:
        _dev_check(t)
        assert t.dtype == torch.float32 and t.is_contiguous()
        rows.append([t.data_ptr(), t.numel(), first])
        first += (t.numel() + 1023) // 1024
    table = torch.from_numpy(np.array(rows, dtype=np.int64)).to(tensors[0].device, non_blocking=True)
    check(lib.wsu_scale_multi_tensor(table.data_ptr(), len(rows), first, factor.data_ptr(), _stream()), "wsu_scale_multi_tensor")


def nonfinite_flag(g: torch.Tensor, flag: torch.Tensor) -> torch.Tensor:
    """flag[0] = 1 if ``g`` holds an inf / NaN else 0, flag[1] += flag[0] (int32[2] on the device; wsu_nonfinite_flag)."""
    lib = _lib.load()
    _dev_check(g, flag)
    assert g.dtype == torch.float32 and g.is_contiguous() and flag.dtype == torch.int32 and flag.numel() >= 2
    check(lib.wsu_nonfinite_flag(g.data_ptr(), g.numel(), flag.data_ptr(), _stream()), "wsu_nonfinite_flag")
    return flag
