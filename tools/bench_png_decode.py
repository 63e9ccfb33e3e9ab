"""Host-only micro-benchmark of libwsu_io's PNG decode (no GPU): ms per 512x512 image on ONE thread and on all threads, over synthetic gray PNGs
written at the compression levels data sets come in, and the reference's real cover (tests/golden/cover_10.png).
python tools/bench_png_decode.py [n_files]"""
import os, sys, time, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from PIL import Image
from ws_unet_amd import formula, _io
from ws_unet_amd.imread import read_luma_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
root = Path(tempfile.mkdtemp())
u8 = formula.synthetic_images(n, 512, 512, seed=99)
sets = {}
for lvl in (1, 6):
    fs = []
    for i in range(n):
        p = root / f"l{lvl}_{i}.png"; Image.fromarray(u8[i]).save(p, compress_level=lvl); fs.append(str(p))
    sets[f"synthetic level {lvl}"] = fs
real = Path(__file__).resolve().parent.parent / "tests" / "golden" / "cover_10.png"
sets["real cover x%d" % n] = [str(real)] * n
out = np.empty((n, 512, 512), np.uint8)
for name, fs in sets.items():
    read_luma_batch(fs, out=out, threads=1)                      # warm the page cache
    ref = np.stack([np.array(Image.open(f).convert("L")) if Image.open(f).mode != "L" else np.array(Image.open(f)) for f in fs[:4]])
    assert np.array_equal(out[:4], ref), name
    t0 = time.perf_counter(); read_luma_batch(fs, out=out, threads=1); t1 = time.perf_counter() - t0
    nt = _io.default_threads()
    t0 = time.perf_counter(); read_luma_batch(fs, out=out, threads=nt); tn = time.perf_counter() - t0
    sz = sum(os.path.getsize(f) for f in fs) / n / 1024
    print(f"{name:24s} {sz:6.0f} KiB/file  1 thread {t1 / n * 1e3:6.3f} ms/image   {nt} threads {tn / n * 1e3:6.3f} ms/image ({n / tn:7.0f} images/s)")
