"""Thin torch-tensor wrappers over the C ABI (include/wsu.h).

PyTorch is plumbing here: it owns device memory (caching allocator) and the HIP
stream; every computation is a libwsu kernel.  Activations are NHWC tensors of
shape (N, H, W, C), fp32 for modes 'f32' / 'bf16x3', bf16 for mode 'bf16'.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import MODES, MODE_BF16, check


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
        self.records.append((kernel, meta, s, e))
        return rc

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


def set_timer(t: Optional[KernelTimer]) -> None:
    global _timer
    _timer = t


def _launch(kernel: str, meta: dict, fn) -> int:
    return fn() if _timer is None else _timer.launch(kernel, meta, fn)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def act_dtype(mode: int) -> torch.dtype:
    return torch.bfloat16 if mode == MODE_BF16 else torch.float32


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
    out = torch.empty(lib.wsu_conv3x3_packed_bytes(cin, cout, mode), dtype=torch.uint8, device=w.device)
    fn = lib.wsu_conv3x3_pack_dgrad if dgrad else lib.wsu_conv3x3_pack
    check(fn(w.data_ptr(), out.data_ptr(), cin, cout, mode, _stream()), "wsu_conv3x3_pack")
    return out


def pack_convt2x2(w: torch.Tensor, mode: int) -> torch.Tensor:
    """w: (Cin, Cout, 2, 2) fp32 on the device -> packed byte buffer."""
    lib = _lib.load()
    w = w.detach()
    _dev_check(w)
    assert w.dtype == torch.float32 and w.dim() == 4 and w.shape[2:] == (2, 2)
    cin, cout = w.shape[:2]
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
    n, h, w, c1 = x1.shape
    c2 = 0 if x2 is None else x2.shape[3]
    dt = act_dtype(mode)
    assert x1.dtype == dt and (x2 is None or (x2.dtype == dt and x2.shape[:3] == x1.shape[:3]))
    y = torch.empty((n, h, w, cout), dtype=dt, device=x1.device)
    yp = idx = None
    if pool:
        yp = torch.empty((n, h // 2, w // 2, cout), dtype=dt, device=x1.device)
        if pool_idx:
            idx = torch.empty((n, h // 2, w // 2, cout), dtype=torch.uint8, device=x1.device)
    esz = 2 if mode == MODE_BF16 else 4
    meta = {"flops": 2.0 * 9 * (c1 + c2) * cout * n * h * w,
            "bytes": float(n * h * w * (c1 + c2 + cout) * esz + (n * h * w // 4 * cout * esz if pool else 0) + 9 * (c1 + c2) * cout * esz)}
    check(_launch("conv3x3", meta, lambda: lib.wsu_conv3x3_fwd(
        x1.data_ptr(), _ptr(x2), w_packed.data_ptr(), _ptr(bias), y.data_ptr(), _ptr(yp), _ptr(idx),
        n, h, w, c1, c2, cout, mode, int(relu), int(pad_zero), _stream())), "wsu_conv3x3_fwd")
    if not pool:
        return y
    return (y, yp, idx) if pool_idx else (y, yp)


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
    n, h, w, cin = x.shape
    y = torch.empty((n, 2 * h, 2 * w, cout), dtype=x.dtype, device=x.device)
    esz = 2 if mode == MODE_BF16 else 4
    meta = {"flops": 2.0 * 4 * cin * cout * n * h * w, "bytes": float(n * h * w * (cin + 4 * cout) * esz + 4 * cin * cout * esz)}
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


def u8_to_unit(x_u8: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _dev_check(x_u8)
    y = torch.empty(x_u8.shape, dtype=torch.float32, device=x_u8.device)
    check(lib.wsu_u8_to_unit_f32(x_u8.data_ptr(), y.data_ptr(), x_u8.numel(), _stream()), "wsu_u8_to_unit_f32")
    return y
