"""ctypes binding of libwsu.so (the C ABI declared in include/wsu.h).

The product path has NO CPU fallback: if the shared library is missing or a call
fails, an exception is raised.  Build with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C ws_unet_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os

# torch MUST be imported before libwsu.so is dlopen'ed: the PyTorch-ROCm wheel bundles its own HIP runtime
# (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  Loaded first, it also satisfies libwsu's DT_NEEDED
# libamdhip64.so.7, so kernels, streams and device pointers live in ONE runtime.  Loaded second, the system
# runtime from libwsu's RUNPATH would be a second, device-less runtime ("no ROCm-capable device").
import torch  # noqa: F401
from ctypes import c_char_p, c_float, c_int, c_longlong, c_size_t, c_uint64, c_void_p
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libwsu.so"

MODE_F32, MODE_BF16X3, MODE_BF16, MODE_BF16X3S, MODE_F16F8, MODE_F16F8X = 0, 1, 2, 3, 4, 5
MODE_F16F8P = 6          # host-side names only: the planar inference path has its own entry points (wsu_*_pl_fwd), no `mode` argument
MODE_F16F8Q = 7          # f16f8p with x_residual = 0 on the first conv of every decoder block
MODE_F16F4P = 8          # planar Q storage; the 3x3 convs multiply their cross terms as block-scaled fp4 (wsu_conv3x3_q_fwd)
MODES = {"f32": MODE_F32, "bf16x3": MODE_BF16X3, "bf16": MODE_BF16, "bf16x3s": MODE_BF16X3S, "f16f8": MODE_F16F8, "f16f8x": MODE_F16F8X, "f16f8p": MODE_F16F8P, "f16f8q": MODE_F16F8Q,
         "f16f4p": MODE_F16F4P}


class WsuError(RuntimeError):
    pass


# name -> (restype, argtypes); must list every symbol of include/wsu.h (tests/test_capi_symbols.py checks)
_P = c_void_p
SIGNATURES = {
    "wsu_version": (c_int, []),
    "wsu_last_error": (c_char_p, []),
    "wsu_act_elem_size": (c_int, [c_int]),
    "wsu_conv3x3_packed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "wsu_conv3x3_pack": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "wsu_conv3x3_pack_dgrad": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "wsu_convt2x2_packed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "wsu_convt2x2_pack": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "wsu_conv3x3_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P] + [c_int] * 9 + [_P]),
    "wsu_conv3x3_head_fwd": (c_int, [_P] * 9 + [c_int] * 8 + [_P]),
    "wsu_conv3x3_fused_first_fwd": (c_int, [_P] * 8 + [c_int] * 6 + [_P]),
    "wsu_conv3x3_wino_packed_bytes": (c_size_t, [c_int, c_int]),
    "wsu_conv3x3_wino_pack": (c_int, [_P, _P, c_int, c_int, _P]),
    "wsu_conv3x3_wino_fwd": (c_int, [_P] * 11 + [c_int] * 8 + [_P]),
    "wsu_conv3x3_pl_fwd": (c_int, [_P] * 10 + [c_int] * 9 + [_P, _P, _P]),
    "wsu_relu_mask_bytes": (c_size_t, [c_int] * 4),
    "wsu_conv3x3_pl_fused_first_fwd": (c_int, [_P] * 7 + [c_int] * 5 + [_P, _P]),
    "wsu_convt2x2_pl_fwd": (c_int, [_P, _P, _P, _P] + [c_int] * 6 + [_P, _P]),
    "wsu_planar_q_bytes": (c_size_t, [c_int] * 4),
    "wsu_conv3x3_q_fwd": (c_int, [_P] * 10 + [c_int] * 9 + [_P, _P]),
    "wsu_conv3x3_packed_f4_bytes": (c_size_t, [c_int] * 2),
    "wsu_conv3x3_pack_f4": (c_int, [_P, _P, c_int, c_int, _P]),
    "wsu_conv3x3_q_fused_first_fwd": (c_int, [_P] * 7 + [c_int] * 5 + [_P, _P]),
    "wsu_conv3x3_up_packed_bytes": (c_size_t, [c_int] * 2),
    "wsu_conv3x3_up_pack": (c_int, [_P] * 7 + [c_int] * 4 + [_P]),
    "wsu_conv3x3_up_q_fwd": (c_int, [_P] * 6 + [c_int] * 7 + [_P, _P]),
    "wsu_conv3x3_pl_bwd_data_workspace_bytes": (c_size_t, [c_int] * 5),
    "wsu_conv3x3_pack_ring": (c_int, [_P, _P, c_int, c_int, _P]),
    "wsu_conv3x3_pl_bwd_data": (c_int, [_P, _P, _P, _P, c_size_t, _P, _P, c_int, _P, _P, _P, _P] + [c_int] * 7 + [_P]),
    "wsu_conv3x3_pl_bwd_weight": (c_int, [_P] * 6 + [c_size_t] + [c_int] * 7 + [_P]),
    "wsu_convt2x2_pl_bwd_weight": (c_int, [_P] * 5 + [c_size_t] + [c_int] * 6 + [_P]),
    "wsu_convt2x2_pl_pack_dgrad": (c_int, [_P, _P, c_int, c_int, _P]),
    "wsu_convt2x2_pl_bwd_data": (c_int, [_P] * 4 + [c_int] * 6 + [_P]),
    "wsu_maxpool2x2_pl_bwd": (c_int, [_P] * 4 + [c_int] * 5 + [_P]),
    "wsu_head_pl_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "wsu_conv1x1_sigmoid_pl_bwd": (c_int, [_P] * 8 + [c_size_t] + [c_int] * 6 + [_P]),
    "wsu_chansum_pl_workspace_bytes": (c_size_t, [c_int]),
    "wsu_colsum_pl": (c_int, [_P, _P, _P, c_size_t] + [c_int] * 5 + [_P]),
    "wsu_conv3x3_first_pl_bwd_weight": (c_int, [_P] * 5 + [c_size_t] + [c_int] * 5 + [_P]),
    "wsu_conv3x3_first_pl_bwd_data": (c_int, [_P] * 3 + [c_int] * 6 + [_P]),
    "wsu_conv3x3_first_pl_fwd": (c_int, [_P, _P, _P, _P] + [c_int] * 7 + [_P, _P, _P]),
    "wsu_conv3x3_first_fwd": (c_int, [_P, _P, _P, _P] + [c_int] * 7 + [_P]),
    "wsu_maxpool2x2_fwd": (c_int, [_P, _P, _P] + [c_int] * 5 + [_P]),
    "wsu_convt2x2_fwd": (c_int, [_P, _P, _P, _P] + [c_int] * 6 + [_P]),
    "wsu_conv1x1_sigmoid_fwd": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 6 + [_P]),
    "wsu_uniform_dropout_fwd": (c_int, [_P, _P, _P, _P] + [c_int] * 5 + [c_float, c_uint64, _P]),
    "wsu_ws_residual_stats": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "wsu_ws_attack_workspace_bytes": (c_size_t, [c_int]),
    "wsu_ws_attack": (c_int, [_P, _P, _P, _P, _P, c_int, c_float, c_int, c_int, _P, _P, _P, c_size_t, c_int, c_int, c_int, _P]),
    "wsu_lsb_delta_unit_f32": (c_int, [_P, _P, c_size_t, _P]),
    "wsu_filter3x3_valid_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "wsu_ws_meter_beta": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "wsu_u8_to_unit_f32": (c_int, [_P, _P, c_size_t, _P]),
    # ---- backward / train step
    "wsu_conv3x3_bwd_data_workspace_bytes": (c_size_t, [c_int] * 6),
    "wsu_conv3x3_bwd_data": (c_int, [_P, _P, _P, _P, c_size_t, _P, _P, c_int, _P, _P] + [c_int] * 6 + [_P]),
    "wsu_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "wsu_conv3x3_bwd_weight": (c_int, [_P, _P, _P, _P, _P, _P, c_size_t] + [c_int] * 7 + [_P]),
    "wsu_first_bwd_workspace_bytes": (c_size_t, [c_int] * 5),
    "wsu_conv3x3_first_bwd_weight": (c_int, [_P, _P, _P, _P, _P, c_size_t] + [c_int] * 5 + [_P]),
    "wsu_conv3x3_first_bwd_data": (c_int, [_P, _P, _P] + [c_int] * 5 + [_P]),
    "wsu_convt2x2_bwd_weight": (c_int, [_P, _P, _P, _P, _P, c_size_t] + [c_int] * 6 + [_P]),
    "wsu_convt2x2_packed_dgrad_bytes": (c_size_t, [c_int, c_int, c_int]),
    "wsu_convt2x2_pack_dgrad": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "wsu_convt2x2_bwd_data": (c_int, [_P, _P, _P, _P] + [c_int] * 6 + [_P]),
    "wsu_maxpool2x2_bwd": (c_int, [_P, _P, _P, _P] + [c_int] * 5 + [_P]),
    "wsu_head_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "wsu_conv1x1_sigmoid_bwd": (c_int, [_P] * 8 + [c_size_t] + [c_int] * 6 + [_P]),
    "wsu_l1ws_loss_workspace_bytes": (c_size_t, [c_int]),
    "wsu_l1ws_loss_fwd_bwd": (c_int, [_P] * 9 + [c_size_t, c_int, c_longlong, c_int, c_int, _P]),
    "wsu_adamw_multi_tensor": (c_int, [_P, c_int, c_longlong] + [c_float] * 5 + [c_int, c_float, _P, _P]),
    "wsu_pow2_grad_scale": (c_int, [_P, c_longlong, _P, _P, _P]),
    "wsu_scale_f32": (c_int, [_P, _P, c_longlong, _P, _P]),
    "wsu_scale_multi_tensor": (c_int, [_P, c_int, c_longlong, _P, _P]),
    "wsu_nonfinite_flag": (c_int, [_P, c_longlong, _P, _P]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libwsu.so once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("WSU_LIB", LIB_PATH))
    if not path.exists():
        raise WsuError(
            f"{path} not found: the HIP extension is not built. Run `make -C {_HERE / 'csrc'}` "
            "(or __graft_entry__.build()). There is no CPU fallback for the UNet hot path.")
    lib = ctypes.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.wsu_version() < 100:
        raise WsuError("libwsu.so is older than this Python package; rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().wsu_last_error()
        raise WsuError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
