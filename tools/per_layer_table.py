"""profiles/rNN/per_layer_*.md from a bench.py line: python tools/per_layer_table.py profiles/r03/bench_n1.json.log > profiles/r03/per_layer.md"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
pl = d["roofline"]["per_layer"]
rows = pl["layers"] if isinstance(pl, dict) else pl
print(f"# Per-layer roofline of the predict step (default mode, batch 32 @ 512x512, one MI355X) -- `roofline.per_layer` of `{sys.argv[1].split('/')[-1]}`\n")
print("roof = max(algorithmic FLOPs / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s); bytes = inputs once + outputs once + weights, priced at the format's 3 B per element "
      "AND at SURVEY 8d's 2 B (VERDICT r02 #12).\n")
print("| layer | kernel | ms | GFLOP | MB (3 B) | MB (2 B) | bound | roof / measured (3 B) | roof / measured (2 B) | TFLOP/s | GB/s | tiles (16x32 px x 64 co) | steps per tile |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
g = lambda r, *ks: next((r[k] for k in ks if k in r), None)
f = lambda v, spec: "-" if v is None else format(v, spec)
tot = roof3 = roof2 = 0.0
for r in rows:
    ms = r["ms"]; tot += ms
    fl, b3, b2 = g(r, "gflop", "GFLOP"), g(r, "mbytes", "MB", "mb"), g(r, "mbytes_2B", "MB_2B", "mb_2B")
    f3, f2 = g(r, "frac", "roof_frac"), g(r, "frac_2B", "roof_frac_2B")
    if f3 is not None: roof3 += f3 * ms
    if f2 is not None: roof2 += f2 * ms
    print(f"| {r['layer']} | {g(r, 'kernel')} | {ms:.3f} | {f(fl, '.1f')} | {f(b3, '.1f')} | {f(b2, '.1f')} | {g(r, 'bound')} | {f(f3, '.3f')} | {f(f2, '.3f')} | "
          f"{f(g(r, 'tflops'), '.1f')} | {f(g(r, 'hbm_GBps'), '.1f')} | {g(r, 'tiles') or '-'} | {g(r, 'steps_per_tile') or '-'} |")
ro = d["roofline"]
print(f"\nSum: {tot:.4f} ms measured; {roof3:.4f} ms at the roofs (3 B) -> {roof3 / tot:.4f}; {roof2:.4f} ms (2 B) -> {roof2 / tot:.4f}.  Headline of this run: "
      f"{d['value']:.0f} images/s, {d['ms_per_step']:.2f} ms per step; {ro.get('kernel', '3x3 convs')} {ro['achieved']:.0f} TFLOP/s algorithmic = {ro['frac']:.3f} of 2.5 PFLOP/s"
      + (f"; fused decoder entries ({ro['fused_up']['kernel']}) {ro['fused_up']['achieved']:.0f} TFLOP/s = {ro['fused_up']['frac']:.3f}; both together {ro['all_conv']['frac']:.3f}" if "fused_up" in ro else "")
      + (f"; whole forward {ro['whole_forward']['achieved']:.0f} TFLOP/s = {ro['whole_forward']['frac']:.3f}" if "whole_forward" in ro else "")
      + (f"; transposed convs {ro['convt2x2']['achieved']:.0f} GB/s = {ro['convt2x2']['frac']:.2f} of 8 TB/s" if "convt2x2" in ro else "") + ".\n")
lb = d.get("latency_b1")
if lb:
    print(f"## Batch 1 (`latency_b1`): {lb['gpu_ms_per_image']:.3f} ms GPU time per image, {lb['wall_ms_per_image_synchronised']:.3f} ms synchronised call, "
          f"{lb['wall_ms_per_image_queued']:.3f} ms queued ({lb['images_per_s_queued']:.0f} images/s one image per call)\n")
    print("| layer | us | tiles | CUs occupied | TFLOP/s |\n|---|---|---|---|---|")
    for r in lb["per_layer"]:
        print(f"| {r['layer']} | {r['ms'] * 1e3:.1f} | {r.get('tiles') or '-'} | {f(r.get('cu_occupied'), '.1f') if r.get('cu_occupied') is not None else '-'} | {f(r.get('tflops'), '.1f')} |")
    print("\n(e31 / e32 have 128 tiles at batch 1: the launcher runs them as 256 half-blocks of 32 output channels, kernel variant MSPLIT.)")
