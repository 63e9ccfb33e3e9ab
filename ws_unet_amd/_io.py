"""ctypes binding of libwsu_io.so (include/wsu_io.h): the host-side batched PNG reader of the evaluate loop.
No torch types cross the boundary: file names in, one (N,H,W) uint8 buffer out."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_int, c_void_p
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "libwsu_io.so"
PNG_OK, PNG_IO, PNG_FORMAT, PNG_UNSUPPORTED, PNG_SHAPE = 0, -1, -2, -3, -4

SIGNATURES = {
    "wsu_io_version": (c_int, []),
    "wsu_png_shape": (c_int, [c_char_p, POINTER(c_int), POINTER(c_int)]),
    "wsu_png_read_luma_batch": (c_int, [POINTER(c_char_p), c_int, c_void_p, c_int, c_int, c_int, POINTER(c_int)]),
}

_lib = None


class WsuIoError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("WSU_IO_LIB", LIB_PATH))
    if not path.exists():
        raise WsuIoError(f"{path} not found: build it with `make -C ws_unet_amd/csrc` (or __graft_entry__.build())")
    lib = ctypes.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def usable_cores() -> int:
    """Cores this process may run on: the affinity mask, capped by the cgroup's CPU quota (cpu.max) when one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:                                    # pragma: no cover
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def default_threads() -> int:
    """Decode threads of a batch read: WSU_IO_THREADS, else this RANK's share of the cores the process may run on -- usable cores divided
    by the ranks on this node (LOCAL_WORLD_SIZE, set by torch.distributed.run): eight ranks that each started `usable cores` threads
    oversubscribed a thin host eightfold -- at most 16."""
    env = os.environ.get("WSU_IO_THREADS")
    if env:
        return max(1, int(env))
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    return max(1, min(16, usable_cores() // local_world))
