"""cProfile of the per-image evaluate API (predict_unet_cover: fabrika's python iterator + predict_unet) over synthetic 512x512 PNGs on one GPU:
where the host time of a row goes.  python tools/profile_per_image.py [--images 512]"""
import argparse, cProfile, pstats, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from PIL import Image
from ws_unet_amd import evaluate, formula
from ws_unet_amd.model import get_model
ap = argparse.ArgumentParser(); ap.add_argument("--images", type=int, default=512); a = ap.parse_args()
root = Path(tempfile.mkdtemp()); (root / "images").mkdir()
u8 = formula.synthetic_images(a.images, 512, 512, seed=99)
for i in range(a.images):
    Image.fromarray(u8[i]).save(root / "images" / f"{i}.png", compress_level=1)
(root / "images" / "files.csv").write_text("name,height,width\n" + "".join(f"images/{i}.png,512,512\n" for i in range(a.images)))
m = get_model("unet_2", in_channels=1, out_channels=1, channel=[0], drop_rate=0.)
m.load_state_dict({k: torch.from_numpy(v) for k, v in formula.formula_state_dict(2, "he").items()})
m = m.to("cuda")
evaluate.predict_unet_cover(root, model=m, take_num_images=64, progress_on=False)
torch.cuda.synchronize()
t0 = time.perf_counter(); evaluate.predict_unet_cover(root, model=m, progress_on=False); dt = time.perf_counter() - t0
print(f"per-image API: {a.images / dt:.0f} images/s ({dt / a.images * 1e3:.3f} ms per row)")
pr = cProfile.Profile(); pr.enable(); evaluate.predict_unet_cover(root, model=m, progress_on=False); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
