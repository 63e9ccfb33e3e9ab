#!/usr/bin/env python3
"""Headline benchmark: 512x512 grayscale images/sec of UNet predict (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Workload (`config.workload`): BASELINE.json configs[1] -- `unet_2` forward-only predict on a batch of 32
synthetic 512x512x1 images per GPU, inputs already resident in HBM as the fp32 (N,1,H,W) tensor the
model boundary takes.  One step = one forward pass over one batch.  Multi-GPU = batch sharding: each
rank predicts its own 32 images (weak scaling), no data-path collective; a barrier brackets the timed
region and the slowest rank's time is used.

Precision: default mode 'bf16x3' (bf16 matrix cores, split operands, fp32 accumulate and storage) -- the
fastest mode that meets the 1e-4 MAE gate; the other modes ('bf16', 'f32') are measured in the same run
with fewer steps and reported under `other_modes`.

Extra JSON objects: `roofline` (dominant kernel = conv3x3 implicit GEMM, algorithmic FLOPs / HIP-event
launch time vs the dense bf16 MFMA peak) and `cpu_baseline` (the CPU oracle = torch-CPU restatement of the
reference, batch 1 with autograd on exactly like infere_single, on a bounded sample of the same images).
"""
import argparse
import glob
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np
import torch

PEAK = {"bf16x3": 2.5e15, "bf16x3s": 2.5e15, "f16f8": 2.5e15, "bf16": 2.5e15, "f32": 157.3e12}     # dense MFMA peaks, MI355X_MICROARCH.md
HBM_PEAK = 8.0e12


def build_model(mode, dev):
    from ws_unet_amd import formula
    from ws_unet_amd.model import get_model
    m = get_model("unet_2", in_channels=1, out_channels=1, channel=[0], drop_rate=None, mode=mode)
    sd = formula.formula_state_dict(2, "he")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev)


def timed_steps(model, x, steps, warmup, use_dist, timer=None):
    from ws_unet_amd import ops
    import torch.distributed as dist
    with torch.no_grad():
        for _ in range(warmup):
            model(x)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ops.set_timer(timer)
        t0 = time.perf_counter()
        for _ in range(steps):
            y = model(x)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ops.set_timer(None)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=x.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    return dt, y


def cpu_baseline(sample_u8, budget_s=25.0):
    """Reference-faithful CPU path: batch 1 per call, autograd enabled (infere_single has no no_grad,
    src/unet/evaluate.py:48), fp32, all host threads.  Bounded: stops after `budget_s` seconds."""
    from ws_unet_amd import formula
    from oracle import unet_ref
    # the box's CPU share (cgroup / affinity), not the host's core count: oversubscribing oneDNN's thread
    # pool on a 16-core share of a 256-core host made the baseline 10x slower than it should be
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    ncpu = max(1, min(ncpu, int(os.environ.get("WSU_CPU_THREADS", "16"))))
    torch.set_num_threads(ncpu)
    ref = unet_ref.build_ref(2, formula.formula_state_dict(2, "he"))
    outs, times = [], []
    x_all = torch.from_numpy(sample_u8.astype(np.float32) / np.float32(255.))[:, None]
    ref(x_all[:1].clone())                                      # warm-up (oneDNN primitive creation)
    t_start = time.perf_counter()
    for i in range(x_all.shape[0]):
        t0 = time.perf_counter()
        y = ref(x_all[i:i + 1].clone())
        times.append(time.perf_counter() - t0)
        outs.append(y.detach())
        if time.perf_counter() - t_start > budget_s:
            break
    return torch.cat(outs), float(np.median(times)), len(times), torch.get_num_threads()


def pmc_traffic(args, algorithmic_bytes_per_launch):
    """HBM bytes per conv3x3 launch from the PMC counters.  They cannot be read from inside this process (rocprofv3 has to
    wrap it), so the number comes from the committed summary of two `rocprofv3 --pmc` passes of THIS command (FETCH_SIZE and
    WRITE_SIZE separately, FETCH_SIZE doubled on gfx950; tools/pmc_traffic.py), and only when workload and mode match."""
    out = {"traffic": None, "algorithmic_bytes_per_launch": algorithmic_bytes_per_launch}
    if (args.batch, args.size) != (32, 512):
        return out
    here = os.path.dirname(os.path.abspath(__file__))
    cands = sorted(glob.glob(os.path.join(here, "profiles", "r*", "pmc_conv3x3_traffic.json")))
    if not cands:
        return out
    with open(cands[-1]) as f:
        pmc = json.load(f)
    if pmc.get("mode", "bf16x3") != args.mode:
        return out
    out["traffic"] = pmc["traffic_bytes_per_launch"]
    out["traffic_unit"] = "HBM bytes per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE)"
    out["traffic_source"] = os.path.relpath(cands[-1], here)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default=os.environ.get("WSU_BENCH_MODE", "f16f8"), choices=["bf16x3", "bf16x3s", "f16f8", "bf16", "f32"])
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-modes", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # Under torch.distributed.run (RANK set) the RCCL process group is always created, also for one rank, so the
    # barrier / max-reduce code below is the same code at every N.
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from ws_unet_amd import formula, ops
    # each rank gets its own shard of the synthetic image stream (global index = rank*batch + i)
    u8 = formula.synthetic_images(args.batch, args.size, args.size, seed=1000 + rank)
    x = ops.u8_to_unit(torch.from_numpy(u8).to(dev))[:, None].contiguous()        # (B,1,H,W) fp32 resident in HBM

    model = build_model(args.mode, dev)
    timer = ops.KernelTimer()
    dt, y = timed_steps(model, x, args.steps, args.warmup, use_dist, timer)
    imgs = world * args.batch * args.steps
    value = imgs / dt

    result = None
    if rank == 0:
        ks = timer.summary()
        conv = ks["conv3x3"]
        achieved = conv["flops"] / (conv["total_ms"] * 1e-3)
        roofline = {
            "bound": "mfma", "kernel": "conv3x3_kernel", "achieved": achieved / 1e12, "peak": PEAK[args.mode] / 1e12,
            "unit": "TFLOP/s", "frac": achieved / PEAK[args.mode], "traffic": None,
            "avg_launch_ms": conv["avg_ms"], "launches": conv["launches"],
            "algorithmic_gflop_per_launch": conv["flops"] / conv["launches"] / 1e9,
            "hbm_GBps_algorithmic": conv["bytes"] / (conv["total_ms"] * 1e-3) / 1e9,
            "note": "algorithmic FLOPs = 2*9*Cin*Cout*N*H*W summed over the 9 conv3x3 launches of a forward "
                    "(the split modes issue extra MFMAs per product -- bf16x3: 3 bf16; f16f8: 1 f16 + one fp8 instruction per tap pair -- they are not counted)",
        }
        # matrix-pipe occupancy next to the algorithmic fraction: one unit = one 32x32x16 bf16/f16 MFMA (32 cycles, 32 768 FLOP); per
        # 16-channel chunk and 32x32 output tile a product costs 9 units of work, the modes issue 27 (bf16x3: 3 per tap), 19 (f16f8: 9
        # f16 + 5 fp8 instructions of 2 units), 9 (bf16) -- the fused first layer and the 16x16x32 shape issue the same unit counts
        units = {"bf16x3": 27 / 9, "bf16x3s": 27 / 9, "f16f8": 19 / 9, "bf16": 1.0}.get(args.mode)
        if units is not None:
            roofline["mfma_issue"] = {"units_per_product": units, "tflops_equivalent": achieved * units / 1e12,
                                      "frac_of_peak": achieved * units / PEAK[args.mode],
                                      "note": "matrix-pipe time actually issued (split terms included) against the same dense bf16 peak"}
        roofline.update(pmc_traffic(args, conv["bytes"] / conv["launches"]))
        gpu_ms = {k: round(v["total_ms"] / args.steps, 3) for k, v in ks.items()}
        result = {
            "metric": "512x512 grayscale images/sec (UNet predict)", "value": value, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16x3": "bf16x3 (bf16 MFMA on split fp32 operands, fp32 accumulate)",
                      "bf16x3s": "bf16x3 (bf16 MFMA on split fp32 operands, fp32 accumulate; activations stored as their hi/lo halves)",
                      "f16f8": "f16f8 (f16 MFMA on the f16 halves + block-scaled fp8 MFMA on the residual cross terms, fp32 accumulate)",
                      "bf16": "bf16", "f32": "f32"}[args.mode],
            "data": "synthetic",
            "config": {"workload": f"unet_2 forward-only predict, batch={args.batch}/GPU synthetic {args.size}x{args.size}x1 "
                                   "(BASELINE.json configs[1]), formula 'he' weights, inputs resident in HBM",
                       "mode": args.mode, "global_batch": world * args.batch, "parallelism": f"batch-shard x{world}"},
            "roofline": roofline,
            "kernel_ms_per_step": gpu_ms,
        }

    # other precision modes, same run, fewer steps (rank 0 only, N = 1 only)
    if rank == 0 and world == 1 and not args.no_other_modes:
        other = {}
        for md in [m for m in ("bf16", "f32", "bf16x3", "bf16x3s", "f16f8") if m != args.mode]:
            mm = build_model(md, dev)
            st = max(2, args.steps // 3)
            d2, y2 = timed_steps(mm, x, st, 1, False)
            other[md] = {"images_per_s": args.batch * st / d2, "ms_per_step": d2 / st * 1e3, "_y": y2[:4].cpu()}
            del mm
        result["other_modes"] = other

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        nsample = 16
        ref_y, t_med, n_done, threads = cpu_baseline(u8[:nsample])
        mae = (y[:n_done].cpu() - ref_y).abs().mean().item()
        result["cpu_baseline"] = {
            "value": 1.0 / t_med, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{n_done} of the same 512x512 images, batch 1 per call, autograd on (reference-faithful "
                      f"infere_single), torch-CPU fp32 restatement of the reference (oracle/unet_ref.py); median {t_med:.3f} s/image",
        }
        result["mae_vs_cpu_oracle"] = mae
        result["speedup_vs_cpu"] = value / (1.0 / t_med)
        if "other_modes" in result:
            for md, o in result["other_modes"].items():
                k = min(4, n_done)
                o["mae_vs_cpu_oracle"] = (o.pop("_y")[:k] - ref_y[:k]).abs().mean().item()
    if rank == 0:
        for o in result.get("other_modes", {}).values():
            o.pop("_y", None)
        print(json.dumps(result))
    if use_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
