"""End-to-end evaluate loop on one GPU (BASELINE.json configs[3] shape, single rank): a synthetic data set of 512x512 gray PNGs on disk,
`predict_unet_cover` (reference-faithful per-image path) vs `predict_unet_cover_batched` (u8 upload -> forward -> fused WS statistics).
Prints one JSON line.  Usage: python tools/bench_evaluate.py [--images 256] [--batch 32] [--mode bf16x3]"""
import argparse, json, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from PIL import Image
from ws_unet_amd import evaluate, fabrika, formula
from ws_unet_amd.model import get_model

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=256)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--mode", default="f16f4p")
a = ap.parse_args()
root = Path(tempfile.mkdtemp())
(root / "images").mkdir()
u8 = formula.synthetic_images(a.images, 512, 512, seed=99)
t0 = time.perf_counter()
for i in range(a.images):
    Image.fromarray(u8[i]).save(root / "images" / f"{i}.png", compress_level=1)
(root / "images" / "files.csv").write_text("name,height,width\n" + "".join(f"images/{i}.png,512,512\n" for i in range(a.images)))
t_write = time.perf_counter() - t0
m = get_model("unet_2", in_channels=1, out_channels=1, channel=[0], drop_rate=0., mode=a.mode)
m.load_state_dict({k: torch.from_numpy(v) for k, v in formula.formula_state_dict(2, "he").items()})
m = m.to("cuda")
batched = fabrika.precovers(iterator="batched", convert_to="pandas", ignore_missing=False, batch_size=a.batch)(
    evaluate._drop_model_kw(evaluate.predict_unet_batch))
batched(root, model=m, take_num_images=a.batch)                 # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter(); dfb = batched(root, model=m); torch.cuda.synchronize(); t_b = time.perf_counter() - t0
n1 = min(a.images, 48)
t0 = time.perf_counter(); df1 = evaluate.predict_unet_cover(root, model=m, take_num_images=n1); torch.cuda.synchronize(); t_1 = time.perf_counter() - t0
t0 = time.perf_counter()
for i in range(min(a.images, 64)):
    np.array(Image.open(root / "images" / f"{i}.png"))
t_dec = (time.perf_counter() - t0) / min(a.images, 64)
err = float(np.abs(dfb["beta_hat"].to_numpy(float)[:n1] - df1["beta_hat"].to_numpy(float)).max())
print(json.dumps({"metric": "evaluate loop images/s (PNG on disk -> beta_hat, l1)", "mode": a.mode, "images": a.images,
                  "batched_images_per_s": a.images / t_b, "per_image_api_images_per_s": n1 / t_1,
                  "png_decode_ms_per_image_1thread": t_dec * 1e3, "max_abs_beta_diff_batched_vs_per_image": err,
                  "note": "batched path: PNG decode by libwsu_io on C++ threads, one chunk ahead of the GPU (fabrika iterator='batched' prefetch); the per-image API decodes with PIL on one thread"}))
