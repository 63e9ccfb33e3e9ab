"""CPU tests of the host-side mirror of the reference interface: fabrika iterator (against golden rows
produced by the reference's fabrika), input transform, image reading."""
import json
from pathlib import Path

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import GOLDEN
from ws_unet_amd import fabrika
from ws_unet_amd.data import get_timm_transform, center_crop, to_tensor
from ws_unet_amd.imread import imread4_u8, imread4_f32, imread_u8
from oracle import evaluate_ref


def echo(fname, **kw):
    return {**kw, "fname": str(fname)}


cover_it = fabrika.precovers(iterator="python", convert_to="pandas", ignore_missing=False, n_jobs=-1)(echo)
stego_it = fabrika.stego_spatial(iterator="python", convert_to="pandas", ignore_missing=False, n_jobs=-1)(echo)


def make_syn(tmp: Path):
    """Same layout as tests/golden/make_golden.py:gen_fabrika builds for the reference."""
    (tmp / "images").mkdir()
    names = [f"images/{i}.png" for i in (1, 10, 11, 2, 20, 3)]
    (tmp / "images" / "files.csv").write_text("name,height,width\n" + "".join(f"{n},512,512\n" for n in names))
    (tmp / "images_b").mkdir()
    (tmp / "images_b" / "files.csv").write_text("name,height,width\nimages_b/7.png,256,256\n")
    sd = tmp / "stego_X_alpha_0.4"
    sd.mkdir()
    sd.joinpath("files.csv").write_text("name,height,width,stego_method,alpha\n" + "".join(
        f"stego_X_alpha_0.4/{i}.png,512,512,X,0.4\n" for i in (1, 10, 2)))


def rows(df, root):
    out = json.loads(df.to_json(orient="records"))
    for r in out:
        r["fname"] = r["fname"].replace(str(root), "<DATASET>")
    return out


@pytest.fixture(scope="module")
def gold():
    return json.loads((GOLDEN / "fabrika.json").read_text())


def test_fabrika_matches_reference_rows(tmp_path, gold):
    make_syn(tmp_path)
    assert rows(cover_it(tmp_path), tmp_path) == gold["syn_covers"]
    assert rows(cover_it(tmp_path, take_num_images=2), tmp_path) == gold["syn_covers_take2"]
    assert rows(cover_it(tmp_path, shuffle_seed=5), tmp_path) == gold["syn_covers_shuffle5"]
    assert rows(stego_it(tmp_path, stego_method="X", alpha=0.4), tmp_path) == gold["syn_stego"]
    # lexical order: images/1, images/10, images/11, images/2 ...
    assert [r["name"] for r in gold["syn_covers"]][:4] == ["images/1.png", "images/10.png", "images/11.png", "images/2.png"]
    with pytest.raises(Exception, match="pre_fn\\(\\) returned empty dataframe"):
        stego_it(tmp_path, stego_method="nope")
    assert gold["syn_stego_empty_error"] == "pre_fn() returned empty dataframe"


def test_fabrika_reference_dataset_rows_recorded(gold):
    """What the reference's own data dir yields (recorded, the dir itself does not travel): 5 covers in
    lexical order, one cover in split_te.csv, 30 LSBR stego rows."""
    assert [r["name"] for r in gold["ref_covers"]] == [f"images/{i}.png" for i in (10, 6, 7, 8, 9)]
    assert [r["name"] for r in gold["ref_covers_split_te"]] == ["images/10.png"]
    assert len(gold["ref_stego_lsbr"]) == 30 and len(gold["ref_stego_lsbr_04"]) == 5


def test_fabrika_iterators_and_errors(tmp_path):
    make_syn(tmp_path)
    batched = fabrika.precovers(iterator="batched", convert_to="pandas", batch_size=4)(
        lambda fnames, kws: [{**kw, "fname": str(f)} for f, kw in zip(fnames, kws)])
    pd.testing.assert_frame_equal(batched(tmp_path), cover_it(tmp_path))
    # the batched iterator builds its rows from records per chunk: the same native Python objects the per-row path passes to fn
    seen = {}
    fabrika.precovers(iterator="python")(lambda f, **kw: seen.setdefault("py", []).append((type(f), {k: type(v) for k, v in kw.items()})))(tmp_path)
    fabrika.precovers(iterator="batched", batch_size=4)(
        lambda fnames, kws: [seen.setdefault("b", []).append((type(f), {k: type(v) for k, v in kw.items()})) for f, kw in zip(fnames, kws)])(tmp_path)
    assert seen["b"] == seen["py"]
    as_np = fabrika.precovers(iterator="python", convert_to="numpy")(lambda f, **kw: kw["height"])
    assert as_np(tmp_path).tolist() == [512] * 6 + [256]
    whole = fabrika.precovers(iterator=None, convert_to=None)(lambda df, **kw: list(df["name"]))
    assert whole(tmp_path)[0] == str(tmp_path / "images/1.png")
    with pytest.raises(NotImplementedError, match="unknown iterator"):
        fabrika.precovers(iterator="spark")(echo)(tmp_path)
    with pytest.raises(NotImplementedError, match="unknown convertor"):
        fabrika.precovers(convert_to="arrow")(echo)(tmp_path)
    jl = fabrika.precovers(iterator="joblib", convert_to="pandas", n_jobs=2)(echo)
    pd.testing.assert_frame_equal(jl(tmp_path), cover_it(tmp_path))
    # extra kwargs flow to fn merged over the row (fabrika.py:86-89)
    df = cover_it(tmp_path, tag="t1", take_num_images=1)
    assert df["tag"].tolist() == ["t1"]
    assert fabrika.filename_to_image_seed("a/b/10.png") == fabrika.filename_to_image_seed("10.jpg")
    assert 0 <= fabrika.filename_to_image_seed("x") < 2 ** 31


def test_cover_stego_pairs(tmp_path):
    make_syn(tmp_path)
    # cover rows need a stego_method column to be split; emulate a split csv carrying both kinds
    rows_ = ["name,height,width,stego_method,alpha"]
    rows_ += [f"images/{i}.png,512,512,," for i in (1, 10, 2)]
    rows_ += [f"stego_X_alpha_0.4/{i}.png,512,512,X,0.4" for i in (1, 10)]
    (tmp_path / "split.csv").write_text("\n".join(rows_) + "\n")
    pairs = fabrika.cover_stego_spatial(iterator="python", convert_to="pandas")(echo)(tmp_path, split="split.csv", stego_method="X")
    assert pairs["name_c"].tolist() == ["images/1.png", "images/10.png", "images/2.png"]
    assert pairs["name_s"].tolist()[:2] == ["stego_X_alpha_0.4/1.png", "stego_X_alpha_0.4/10.png"]
    assert pd.isna(pairs["name_s"].tolist()[2])


def test_transform_semantics():
    tf = get_timm_transform(mean=None, std=None, grayscale=True)
    x = np.random.default_rng(0).integers(0, 256, (512, 512, 1)).astype(np.float32)
    t = tf(x / 255.)
    assert t.shape == (1, 512, 512) and t.dtype == torch.float32
    np.testing.assert_array_equal(t.numpy()[0], (x / 255.)[..., 0])            # identity on 512x512 gray
    assert torch.equal(t, evaluate_ref.transform_gray((x / 255.).astype(np.float32)))
    # 4-plane input keeps plane 3; larger / smaller images are centre-cropped / zero-padded
    x4 = np.random.default_rng(1).random((600, 520, 4)).astype(np.float32)
    t4 = tf(x4)
    assert torch.equal(t4, evaluate_ref.transform_gray(x4))
    np.testing.assert_array_equal(t4.numpy()[0], x4[44:556, 4:516, 3])
    small = np.ones((100, 101, 1), np.float32)
    ts = tf(small)
    assert ts.shape == (1, 512, 512) and ts.sum().item() == 100 * 101
    assert torch.equal(ts, evaluate_ref.transform_gray(small))
    assert ts[0, 206, 205].item() == 1 and ts[0, 205, 205].item() == 0 and ts[0, 206, 204].item() == 0
    # uint8 ToTensor scales, float does not
    assert to_tensor(np.full((2, 2, 1), 255, np.uint8)).max().item() == 1.0
    # optional oracles append planes
    tp = get_timm_transform(None, None, grayscale=True, parity_oracle=True, demosaic_oracle=True)(x / 255.)
    assert tp.shape == (5, 512, 512)
    np.testing.assert_array_equal(tp[1].numpy(), (x[..., 0].astype(np.int64) & 1).astype(np.float32))
    assert tp[2, 0, 0] == 1 and tp[3, 0, 1] == 1 and tp[3, 1, 0] == 1 and tp[4, 1, 1] == 1


def test_imread_gray_png():
    x4 = imread4_u8(GOLDEN / "cover_10.png")
    assert x4.shape == (512, 512, 4) and x4.dtype == np.uint8
    for c in range(3):
        np.testing.assert_array_equal(x4[..., c], x4[..., 3])
    np.testing.assert_array_equal(x4[..., 3:], imread_u8(GOLDEN / "cover_10.png"))
    assert imread4_f32(GOLDEN / "cover_10.png").dtype == np.float32


def test_native_png_reader_matches_pil(tmp_path):
    """libwsu_io (include/wsu_io.h): every PNG filter type, gray and RGB, vs PIL / the imread4_u8 luma; fallbacks and errors."""
    import ctypes
    from PIL import Image
    from ws_unet_amd import _io, formula
    from ws_unet_amd.imread import imread4_u8, png_shape, read_luma_batch
    lib = _io.load()
    assert lib.wsu_io_version() == 100
    text = (Path(__file__).resolve().parent.parent / "include" / "wsu_io.h").read_text()
    import re
    decl = sorted(set(re.findall(r"\b(wsu_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", text, flags=re.S))))
    assert decl == sorted(_io.SIGNATURES) and all(hasattr(lib, s) for s in decl)
    u8 = formula.synthetic_images(3, 40, 56, seed=11)
    files = []
    for i, (img, kw) in enumerate([(u8[0], {}), (u8[1], {"compress_level": 1}), (u8[2], {"optimize": True})]):
        f = tmp_path / f"g{i}.png"
        Image.fromarray(img).save(f, **kw)
        files.append(f)
    # a noisy and a smooth RGB image (different filter choices inside the encoder)
    rgb = np.stack([u8[0], u8[1], u8[2]], axis=-1)
    Image.fromarray(rgb).save(tmp_path / "c0.png")
    grad = (np.add.outer(np.arange(40), np.arange(56)) % 256).astype(np.uint8)
    Image.fromarray(np.stack([grad, grad[::-1], grad.T[:40, :56] if grad.T.shape == (40, 56) else grad], axis=-1)).save(tmp_path / "c1.png")
    files += [tmp_path / "c0.png", tmp_path / "c1.png"]
    assert png_shape(files[0]) == (40, 56)
    out = read_luma_batch(files, threads=3)
    for i, f in enumerate(files):
        np.testing.assert_array_equal(out[i], imread4_u8(f)[..., 3], err_msg=str(f))
    np.testing.assert_array_equal(read_luma_batch([GOLDEN / "cover_10.png"])[0], imread4_u8(GOLDEN / "cover_10.png")[..., 3])
    # unsupported variant (palette) falls back to PIL for that file only
    Image.fromarray(u8[0]).convert("P").save(tmp_path / "p.png")
    out2 = read_luma_batch([files[0], tmp_path / "p.png"])
    np.testing.assert_array_equal(out2[1], imread4_u8(tmp_path / "p.png")[..., 3])
    # errors: ragged shapes, missing file
    Image.fromarray(u8[0][:20]).save(tmp_path / "small.png")
    with pytest.raises(ValueError, match="shape"):
        read_luma_batch([files[0], tmp_path / "small.png"])
    with pytest.raises(OSError):
        read_luma_batch([files[0], tmp_path / "missing.png"])
    assert read_luma_batch([]).shape[0] == 0


def _png_bytes(img: np.ndarray, filters, idat_chunk: int = 0, level: int = 6) -> bytes:
    """A PNG of `img` ((H,W) gray or (H,W,3) RGB uint8) whose scanline y is encoded with filter type filters[y % len(filters)] (test-side
    encoder, PNG spec section 9), the zlib stream cut into IDAT chunks of `idat_chunk` bytes (0: one chunk)."""
    import struct
    import zlib
    h, w = img.shape[:2]
    bpp = 1 if img.ndim == 2 else 3
    rows = img.reshape(h, w * bpp).astype(np.int32)
    raw = bytearray()
    prev = np.zeros(w * bpp, np.int32)
    for y in range(h):
        cur = rows[y]
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        ul = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        ft = filters[y % len(filters)]
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - left
        elif ft == 2:
            f = cur - prev
        elif ft == 3:
            f = cur - ((left + prev) >> 1)
        else:
            p_ = left + prev - ul
            pa, pb, pc = np.abs(p_ - left), np.abs(p_ - prev), np.abs(p_ - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            f = cur - pred
        raw.append(ft)
        raw += bytes((f & 255).astype(np.uint8))
        prev = cur
    z = zlib.compress(bytes(raw), level)
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0 if bpp == 1 else 2, 0, 0, 0))
    step = idat_chunk or len(z)
    for k in range(0, len(z), step):
        out += chunk(b"IDAT", z[k:k + step])
    return out + chunk(b"IEND", b"")


def test_native_png_reader_every_filter_type_and_stream_shape(tmp_path):
    """The streaming reader of round 4 (inflate through a 16-row window, scanlines unfiltered straight into the destination, SSE2 Sub): every
    filter type on every row position (first row included), widths around the 16-byte vector step, zlib streams in one and in many IDAT chunks
    (also cut inside a scanline), gray and RGB -- bit-exact against the source image; truncated / corrupt streams are errors, not garbage."""
    from ws_unet_amd.imread import png_shape, read_luma_batch
    rng = np.random.default_rng(4)
    for w in (1, 15, 16, 17, 56, 130):
        for h in (1, 5, 37):
            gray = rng.integers(0, 256, (h, w), dtype=np.uint8)
            smooth = ((np.add.outer(np.arange(h), np.arange(w)) * 3) % 256).astype(np.uint8)
            for img in (gray, smooth):
                for filters, cut in (([0], 0), ([1], 0), ([2], 0), ([3], 0), ([4], 0), ([4, 1, 3, 2, 0], 0), ([1, 4, 2, 3], 7), ([3, 4], 100)):
                    f = tmp_path / "t.png"
                    f.write_bytes(_png_bytes(img, filters, cut))
                    assert png_shape(f) == (h, w)
                    np.testing.assert_array_equal(read_luma_batch([f])[0], img, err_msg=f"gray {w}x{h} filters {filters} cut {cut}")
    rgb = rng.integers(0, 256, (23, 41, 3), dtype=np.uint8)
    luma = ((rgb[..., 0].astype(np.int64) * 4899 + rgb[..., 1].astype(np.int64) * 9617 + rgb[..., 2].astype(np.int64) * 1868 + (1 << 13)) >> 14).astype(np.uint8)
    for filters, cut in (([0, 1, 2, 3, 4], 0), ([4], 33), ([3, 1], 5)):
        f = tmp_path / "c.png"
        f.write_bytes(_png_bytes(rgb, filters, cut))
        np.testing.assert_array_equal(read_luma_batch([f])[0], luma)
    # a batch larger than the thread count: every thread reuses its scratch for several files of different sizes of zlib stream
    files = []
    imgs = rng.integers(0, 256, (9, 64, 80), dtype=np.uint8)
    for i in range(9):
        files.append(tmp_path / f"b{i}.png")
        files[-1].write_bytes(_png_bytes(imgs[i], [i % 5, 4], (0, 50, 4096)[i % 3], level=(1, 6, 9)[i % 3]))
    np.testing.assert_array_equal(read_luma_batch(files, threads=2), imgs)
    # corrupt: truncated zlib stream, extra scanlines, a bad filter byte
    good = _png_bytes(imgs[0], [4])
    bad = tmp_path / "bad.png"
    bad.write_bytes(good[:len(good) // 2])
    with pytest.raises(OSError):
        read_luma_batch([bad])
    bad.write_bytes(_png_bytes(np.concatenate([imgs[0], imgs[1]]), [1]).replace(b"\x00\x00\x00\x80", b"\x00\x00\x00\x40", 1))   # IHDR says 64 rows, the stream holds 128 (CRCs are not checked, like cv2's fast path)
    with pytest.raises(OSError):
        read_luma_batch([bad])
    bad.write_bytes(_png_bytes(imgs[0], [7]))
    with pytest.raises(OSError):
        read_luma_batch([bad])


def test_u8_shards_serve_the_same_planes(tmp_path):
    """Pre-decoded uint8 shards (evaluate.write_u8_shards / use_u8_shards, SURVEY 8d ".npy covers"): load_planes_u8 returns the bytes of the PNG
    decode; a file rewritten after the shards were made is decoded again instead of served stale; unknown files fall back to the decode."""
    from PIL import Image
    from ws_unet_amd import evaluate, formula
    u8 = formula.synthetic_images(5, 64, 48, seed=8)
    files = []
    for i in range(5):
        files.append(tmp_path / f"{i}.png")
        Image.fromarray(u8[i]).save(files[-1])
    try:
        sd = evaluate.write_u8_shards(files[:4], tmp_path / "shards", images_per_shard=3)
        assert sorted(p.name for p in sd.iterdir()) == ["index.json", "planes_0000.npy", "planes_0001.npy"]
        assert evaluate.use_u8_shards(sd) == 4
        buf = np.zeros((2, 64, 48), np.uint8)
        assert evaluate._planes_from_shards([str(files[2]), str(files[3])], buf) and np.array_equal(buf, u8[2:4])
        np.testing.assert_array_equal(evaluate.load_planes_u8([str(f) for f in files[:4]]).numpy(), u8[:4])
        np.testing.assert_array_equal(evaluate.load_planes_u8([str(files[3]), str(files[4])]).numpy(), u8[3:5])      # one file not indexed: decoded
        assert not evaluate._planes_from_shards([str(files[3]), str(files[4])], buf)
        import os, time
        Image.fromarray(u8[4]).save(files[0])                                     # file 0 now holds image 4
        os.utime(files[0], ns=(time.time_ns(), time.time_ns() + 5_000_000))
        assert not evaluate._planes_from_shards([str(files[0])], buf[:1])
        np.testing.assert_array_equal(evaluate.load_planes_u8([str(files[0])]).numpy()[0], u8[4])
    finally:
        evaluate.use_u8_shards(None)
    b = evaluate.decode_budget([str(f) for f in files[1:]], gpu_images_per_s=3000.0)
    assert b["decode_ms_per_image_per_thread"] > 0 and b["threads_needed_per_rank"] > 0 and b["usable_cores"] >= 1


@pytest.mark.parametrize("depth", [None, 3])
def test_python_iterator_announces_the_next_file(tmp_path, depth):
    """iterator='python': a fn with a `lookahead` attribute is told row i + 1's file before it runs on row i (the per-image evaluate API decodes
    it on a helper thread meanwhile); rows, order and results are those of the plain loop."""
    make_syn(tmp_path)
    log = []

    def fn(fname, **kw):
        log.append(("run", Path(fname).name))
        return {**kw, "fname": str(fname)}
    fn.lookahead = lambda f: log.append(("ahead", Path(f).name))
    if depth:
        fn.lookahead_depth = depth                                             # rows i + 1 .. i + depth are announced before row i runs
    df = fabrika.precovers(iterator="python", convert_to="pandas", ignore_missing=False)(fn)(tmp_path)
    pd.testing.assert_frame_equal(df, cover_it(tmp_path))
    runs = [n for k, n in log if k == "run"]
    assert [n for k, n in log if k == "ahead"] == runs[1:]                       # every file but the first, in order
    for i, name in enumerate(runs[1:]):
        assert log.index(("ahead", name)) < log.index(("run", runs[max(0, i + 1 - (depth or 1))]))   # announced before the row `depth` in front of it runs


def test_batched_iterator_prefetch_runs_one_chunk_ahead(tmp_path):
    (tmp_path / "images").mkdir()
    names = [f"images/{i}.png" for i in range(7)]
    (tmp_path / "images" / "files.csv").write_text("name,height,width\n" + "".join(f"{n},8,8\n" for n in names))
    log = []

    def fn(fnames, kws, prefetched=None):
        log.append(("run", [Path(f).name for f in fnames], prefetched))
        return [{**kw, "tag": prefetched} for kw in kws]

    def prefetch(fnames, kws):
        log.append(("pre", [Path(f).name for f in fnames]))
        return "staged:" + Path(fnames[0]).name

    fn.prefetch = prefetch
    df = fabrika.precovers(iterator="batched", convert_to="pandas", ignore_missing=False, batch_size=3)(fn)(tmp_path)
    assert df["name"].tolist() == sorted(names) and len(df) == 7
    assert df["tag"].tolist() == ["staged:0.png"] * 3 + ["staged:3.png"] * 3 + ["staged:6.png"]
    runs = [e for e in log if e[0] == "run"]
    assert [r[1] for r in runs] == [["0.png", "1.png", "2.png"], ["3.png", "4.png", "5.png"], ["6.png"]]
    # every chunk was staged exactly once, and staging of chunk k+1 was submitted before chunk k ran
    assert [e[1][0] for e in log if e[0] == "pre"] == ["0.png", "3.png", "6.png"]


def test_batched_iterator_submits_the_next_chunk_before_collecting(tmp_path):
    """fn.submit / fn.collect (evaluate.py's split predict): chunk k+1 is launched before chunk k's rows are read back, rows stay in order."""
    (tmp_path / "images").mkdir()
    names = [f"images/{i}.png" for i in range(7)]
    (tmp_path / "images" / "files.csv").write_text("name,height,width\n" + "".join(f"{n},8,8\n" for n in names))
    log = []

    def fn(fnames, kws, prefetched=None):                       # the unsplit form must not be used when the split one exists
        raise AssertionError("unsplit call")

    def submit(fnames, kws, prefetched=None):
        log.append(("submit", Path(fnames[0]).name))
        return (prefetched, kws)

    def collect(handle):
        log.append(("collect", handle[0]))
        return [{**kw, "tag": handle[0]} for kw in handle[1]]

    fn.prefetch = lambda fnames, kws: "staged:" + Path(fnames[0]).name
    fn.submit, fn.collect = submit, collect
    df = fabrika.precovers(iterator="batched", convert_to="pandas", ignore_missing=False, batch_size=3)(fn)(tmp_path)
    assert df["name"].tolist() == sorted(names)
    assert df["tag"].tolist() == ["staged:0.png"] * 3 + ["staged:3.png"] * 3 + ["staged:6.png"]
    assert log == [("submit", "0.png"), ("submit", "3.png"), ("collect", "staged:0.png"), ("submit", "6.png"),
                   ("collect", "staged:3.png"), ("collect", "staged:6.png")]


def _pair_dataset(root, n=10, size=16):
    from PIL import Image
    from ws_unet_amd import formula
    (root / "images").mkdir()
    u8 = formula.synthetic_images(n, size, size, seed=21)
    for i in range(n):
        Image.fromarray(u8[i]).save(root / "images" / f"{i}.png")
    (root / "images" / "files.csv").write_text("name,height,width\n" + "".join(f"images/{i}.png,{size},{size}\n" for i in range(n)))
    sd = root / "stego_LSBR_alpha_0.4"
    sd.mkdir()
    st = {}
    for i in range(n - 1):                                    # cover n-1 has no stego twin
        st[i] = formula.lsbr_embed(u8[i], 0.4, seed=i)
        Image.fromarray(st[i]).save(sd / f"{i}.png")
    (sd / "files.csv").write_text("name,height,width,stego_method,alpha\n" + "".join(
        f"stego_LSBR_alpha_0.4/{i}.png,{size},{size},LSBR,0.4\n" for i in range(n - 1)))
    return u8, st


def test_pair_loader_batches_and_rank_shards(tmp_path):
    from ws_unet_amd.data.pairs import PairLoader
    u8, st = _pair_dataset(tmp_path)
    ld = PairLoader(tmp_path, None, "LSBR", 0.4, batch_size=4, shuffle=False)
    assert len(ld) == 4 and len(ld.covers) == 9               # 9 pairs, 2 per batch, tail dropped
    batches = list(ld)
    assert len(batches) == 4
    x, (c, a) = batches[0]
    assert x.dtype == torch.uint8 and x.shape == (4, 16, 16) and a.tolist() == [0.0, pytest.approx(0.4), 0.0, pytest.approx(0.4)]
    names = sorted(f"images/{i}.png" for i in range(9))        # fabrika's lexical order
    i0, i1 = int(Path(names[0]).stem), int(Path(names[1]).stem)
    np.testing.assert_array_equal(x[0].numpy(), u8[i0]); np.testing.assert_array_equal(x[1].numpy(), st[i0])
    np.testing.assert_array_equal(c[0].numpy(), u8[i0]); np.testing.assert_array_equal(c[1].numpy(), u8[i0])
    np.testing.assert_array_equal(x[3].numpy(), st[i1]); np.testing.assert_array_equal(c[3].numpy(), u8[i1])
    # data-parallel shards: disjoint, equal step counts, union = the single-rank epoch (same seed / epoch)
    full = PairLoader(tmp_path, None, "LSBR", 0.4, batch_size=2, shuffle=True, seed=5)
    r0 = PairLoader(tmp_path, None, "LSBR", 0.4, batch_size=2, shuffle=True, seed=5, rank=0, world=2)
    r1 = PairLoader(tmp_path, None, "LSBR", 0.4, batch_size=2, shuffle=True, seed=5, rank=1, world=2)
    assert len(r0) == len(r1) == 4
    p0, p1 = r0.pair_order().ravel().tolist(), r1.pair_order().ravel().tolist()
    assert not set(p0) & set(p1) and sorted(p0 + p1) == sorted(full.pair_order().ravel().tolist()[:8])
    e0 = full.pair_order().ravel().tolist()
    full.reshuffle()
    assert full.pair_order().ravel().tolist() != e0 and sorted(full.pair_order().ravel().tolist()) == sorted(e0)
    # covers only (the 'dropout' run): one sample per cover, alpha 0, the unpaired cover is kept
    co = PairLoader(tmp_path, None, None, None, batch_size=5, covers_only=True, shuffle=False)
    xb, (cb, ab) = next(iter(co))
    assert len(co.covers) == 10 and torch.equal(xb, cb) and ab.tolist() == [0.0] * 5
    with pytest.raises(ValueError, match="even"):
        PairLoader(tmp_path, None, "LSBR", 0.4, batch_size=3)
    with pytest.raises(Exception):
        PairLoader(tmp_path, None, "HILLR", 0.4, batch_size=2)


def test_gpu_predictor_refuses_pickling_clearly():
    import pickle
    from ws_unet_amd.ws.estimate import UNetEstimator
    with pytest.raises(TypeError, match="cannot be pickled"):
        pickle.dumps(UNetEstimator(model=object()))


def test_train_driver_argument_merge(tmp_path, monkeypatch):
    """`python -m ws_unet_amd.train --config <published config.json> --dataset ...`: config keys are taken over, flags win."""
    from ws_unet_amd import train as train_mod
    cfg = {"network": "unet_2", "alpha": "0.400", "batch_size": 16, "loss": "l1ws", "learning_rate": 0.0001, "drop_rate": 0.0,
           "stego_method": "LSBR", "tr_csv": "split_tr.csv", "va_csv": "split_va.csv", "num_epochs": 300, "patience": 10,
           "dataset": "/gpfs/somewhere", "output_dir": "/gpfs/out", "print_freq": 50, "num_workers": 8, "covers_only": False}
    f = tmp_path / "config.json"
    f.write_text(json.dumps(cfg))
    seen = {}
    monkeypatch.setattr(train_mod, "train", lambda args: seen.update(args) or 0.0)
    train_mod.main(["--config", str(f), "--dataset", str(tmp_path), "--output_dir", str(tmp_path / "runs"), "--num_epochs", "2",
                    "--covers_only", "false", "--channel", "0"])
    assert seen["dataset"] == str(tmp_path) and seen["output_dir"] == str(tmp_path / "runs") and seen["num_epochs"] == 2
    assert seen["network"] == "unet_2" and seen["alpha"] == "0.400" and seen["batch_size"] == 16 and seen["stego_method"] == "LSBR"
    assert "print_freq" not in seen and "num_workers" not in seen          # keys this driver has no use for are dropped
    assert seen["covers_only"] is False and seen["channel"] == [0]
    with pytest.raises(SystemExit):
        train_mod.main([])                                                  # --dataset is mandatory


def test_bench_self_launch_refuses_cleanly_without_gpus():
    """`python bench.py --gpus N` (N > 1, no torch.distributed.run environment) launches the ranks itself; on a box with fewer
    GPUs it must say so and exit non-zero BEFORE touching the GPU (VERDICT r01 missing #1)."""
    import os
    import subprocess
    import sys
    import glob
    if len(glob.glob("/dev/dri/renderD*")) >= 2:
        pytest.skip("box has >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "needs 2 GPUs" in r.stderr and r.stdout.strip() == ""


def test_bench_self_launch_runs_two_ranks_on_cpu():
    """The N > 1 path of `python bench.py --gpus N` end to end without GPUs: the parent spawns torch.distributed.run
    with two ranks, they rendezvous on 127.0.0.1, run the barrier / max-over-ranks code over gloo, and ONLY rank 0's JSON line reaches the
    parent's stdout.  The parent never imports torch (it must stay GPU-free by construction)."""
    import json
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["gloo_world_size"] == 2 and d["steps"] == 3 and "rehearsal" in d["metric"]
    src = (root / "bench.py").read_text()
    body = src[src.index("def self_launch"):src.index("def build_model")]
    assert "import torch" not in body and "visible_gpus" in body


def test_train_products_setting(monkeypatch):
    """`train_products` (include/wsu.h WSU_PRODUCTS_*): default 'f16', the environment overrides, an unknown name is refused when the model is
    built -- not at the first backward."""
    from ws_unet_amd import ops
    from ws_unet_amd.model import get_model
    assert ops.products_id("f16f8") == 0 and ops.products_id("f16") == 1
    with pytest.raises(ValueError, match="products"):
        ops.products_id("bf16")
    monkeypatch.delenv("WSU_TRAIN_PRODUCTS", raising=False)
    m = get_model("unet_0", in_channels=1, out_channels=1, channel=[0], drop_rate=None, mode="f16f8p")
    assert m.train_mode == "f16f8p" and m.train_products == "f16"
    monkeypatch.setenv("WSU_TRAIN_PRODUCTS", "f16f8")
    assert get_model("unet_0", in_channels=1, out_channels=1, channel=[0], drop_rate=None, mode="f16f8p").train_products == "f16f8"
    monkeypatch.setenv("WSU_TRAIN_PRODUCTS", "fp4")
    with pytest.raises(ValueError, match="products"):
        get_model("unet_0", in_channels=1, out_channels=1, channel=[0], drop_rate=None, mode="f16f8p")


def test_per_image_api_rows_ride_along_host_logic(tmp_path, monkeypatch):
    """The host side of the per-image evaluate API without a GPU (the device legs replaced by a stub that returns each plane's own mean / max):
    rows decoded ahead join the launch of the row asked for, one launch is queued ahead, every row gets ITS image's numbers in the reference's
    order, every image is computed once, a file rewritten after it was announced is recomputed, nothing is left behind -- also when the
    predictor raises in the middle of a pass."""
    from PIL import Image
    from ws_unet_amd import evaluate
    n = 30
    rng = np.random.default_rng(3)
    u8 = rng.integers(0, 256, size=(n, 512, 512), dtype=np.uint8)
    (tmp_path / "images").mkdir()
    for i in range(n):
        Image.fromarray(u8[i]).save(tmp_path / "images" / f"{i:02d}.png", compress_level=1)
    (tmp_path / "images" / "files.csv").write_text("name,height,width\n" + "".join(f"images/{i:02d}.png,512,512\n" for i in range(n)))

    class Stub(torch.nn.Module):
        mode = "f32"
        def forward_features(self, x):
            return x
    model = Stub()
    sizes = []

    def fake_batch(x_u8, m):
        sizes.append(int(x_u8.shape[0]))
        f = x_u8.float()
        return f.mean(dim=(1, 2)), f.amax(dim=(1, 2))
    monkeypatch.setattr(evaluate, "predict_u8_batch", fake_batch)
    monkeypatch.setattr(evaluate, "_model_device", lambda m: torch.device("cpu"))
    df = evaluate.predict_unet_cover(tmp_path, model=model, progress_on=False)
    assert df["name"].tolist() == [f"images/{i:02d}.png" for i in range(n)]
    np.testing.assert_allclose(df["beta_hat"].to_numpy(float), u8.reshape(n, -1).mean(axis=1), rtol=1e-6)
    np.testing.assert_array_equal(df["l1"].to_numpy(float), u8.reshape(n, -1).max(axis=1).astype(float))
    assert sum(sizes) == n and max(sizes) <= evaluate._MICRO_BATCH
    assert not evaluate._AHEAD["results"] and not evaluate._AHEAD["pending"] and not evaluate._AHEAD["inflight"]
    # a file rewritten between its announcement and its row
    files = [str(tmp_path / "images" / f"{i:02d}.png") for i in range(4)]
    evaluate._lookahead_reset()
    for f in files[1:]:
        evaluate._lookahead(f)
    for fut, _ in list(evaluate._AHEAD["pending"].values()):
        fut.result()
    r0 = evaluate.predict_unet(files[0], model)
    assert float(r0["beta_hat"]) == pytest.approx(u8[0].mean(), rel=1e-6) and set(evaluate._AHEAD["results"]) == set(files[1:])
    Image.fromarray(u8[9]).save(files[1], compress_level=1)
    r1 = evaluate.predict_unet(files[1], model)
    assert float(r1["beta_hat"]) == pytest.approx(u8[9].mean(), rel=1e-6)
    r2 = evaluate.predict_unet(files[2], model)                      # ... the others come from the cache
    k = len(sizes)
    assert float(r2["beta_hat"]) == pytest.approx(u8[2].mean(), rel=1e-6) and len(sizes) == k
    evaluate._lookahead_reset()
    # a predictor that raises mid-pass: the next pass starts clean
    calls = {"n": 0}

    def boom(x_u8, m):
        calls["n"] += 1
        if calls["n"] == 2:
            raise RuntimeError("device lost")
        return fake_batch(x_u8, m)
    monkeypatch.setattr(evaluate, "predict_u8_batch", boom)
    with pytest.raises(RuntimeError, match="device lost"):
        evaluate.predict_unet_cover(tmp_path, model=model, progress_on=False)
    assert not evaluate._AHEAD["results"] and not evaluate._AHEAD["pending"] and not evaluate._AHEAD["inflight"]
    monkeypatch.setattr(evaluate, "predict_u8_batch", fake_batch)
    df2 = evaluate.predict_unet_cover(tmp_path, model=model, progress_on=False, take_num_images=5)
    np.testing.assert_allclose(df2["beta_hat"].to_numpy(float)[2:], u8.reshape(n, -1).mean(axis=1)[2:5], rtol=1e-6)


def test_forward_organisation_switches_are_read_at_construction(monkeypatch):
    """WSU_FUSE_UP (default on: the decoder blocks' fused entries), WSU_FUSE_FIRST_Q (default off): attributes of the model, so that a caller can A/B them."""
    from ws_unet_amd.model import get_model
    m = get_model("unet_2", in_channels=1, out_channels=1, channel=[0], drop_rate=0.)
    assert m.fuse_up_planar is True and m.fuse_first_q is False and m.mode == "f16f4p"
    monkeypatch.setenv("WSU_FUSE_UP", "0")
    monkeypatch.setenv("WSU_FUSE_FIRST_Q", "1")
    m = get_model("unet_2", in_channels=1, out_channels=1, channel=[0], drop_rate=0.)
    assert m.fuse_up_planar is False and m.fuse_first_q is True
