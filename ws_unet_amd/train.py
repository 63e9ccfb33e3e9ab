"""Training driver for the UNet runs: `python -m ws_unet_amd.train ...` or, data-parallel on one node,
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m ws_unet_amd.train ...`.

The reference publishes no UNet driver; this follows the structure of its detector driver
(src/detector/train.py:143-304: run directory naming, config.json dump, log/ + model/ sub-directories, optional weight-only
resume, epoch loop with reshuffle / validate / checkpoint / best copy / patience) and takes the keys of the published run configs
(models/unet/*/config.json) under the same names, so a published config replays with `--config <file> --dataset <dir>`.

`batch_size` is the GLOBAL batch as in the published configs (16); with N ranks every rank runs batch_size / N samples per step and
the flat gradient bucket is sum-all-reduced over RCCL (parallel.allreduce_flat_).  The UniformDropout mask stream is seeded per
rank (seed * world + rank) and advances with the call counter, so replicas draw different masks and a run is reproducible.
"""
from __future__ import annotations

import argparse
import json
import pathlib
import time
import typing

import torch

from . import parallel
from .data.pairs import PairLoader
from .model import get_model
from .trainer import Trainer, create_run_name, resume

DEFAULTS: typing.Dict[str, typing.Any] = {
    "SLURM_JOB_ID": "0", "alpha": None, "batch_size": 16, "channel": [0], "covers_only": False, "dataset": None,
    "demosaic": None, "demosaic_oracle": False, "drop_rate": 0.0, "experiment_dir_suffix": None, "grayscale": True,
    "learning_rate": 1e-4, "loss": "l1ws", "loss_lambda": 0.25, "network": "unet_2", "num_epochs": 300, "output_dir": "runs",
    "patience": 10, "resume": None, "resume_dir": None, "seed": None, "stego_method": None, "tr_csv": "split_tr.csv",
    "va_csv": "split_va.csv", "take_num_images": None, "mode": None, "train_mode": None,
}


def train(args: typing.Dict[str, typing.Any]) -> float:
    """One run; returns the best validation loss.  `args`: DEFAULTS overridden by the caller (published config keys)."""
    args = {**DEFAULTS, **args}
    rank, world = parallel.init_from_env()
    if not torch.cuda.is_available():
        raise RuntimeError("ws_unet_amd.train needs a GPU: the train step is libwsu kernels only (no CPU path)")
    dev = torch.device("cuda", torch.cuda.current_device())
    if args["batch_size"] % world:
        raise ValueError(f"global batch_size {args['batch_size']} is not divisible by {world} ranks")

    # run directory: <output_dir>/<stego_method>/<yymmddHHMMSS>-<job>-<run name>[_suffix]   (detector/train.py:146-159)
    name = time.strftime("%y%m%d%H%M%S") + "-" + str(args["SLURM_JOB_ID"]) + "-" + create_run_name(args)
    if args["experiment_dir_suffix"]:
        name += "_" + args["experiment_dir_suffix"]
    method_dir = args["stego_method"] or "dropout"
    out_dir = pathlib.Path(args["output_dir"]) / method_dir / name
    if world > 1:                                                        # every rank must agree on the timestamped name
        box = [str(out_dir)]
        torch.distributed.broadcast_object_list(box, src=0)
        out_dir = pathlib.Path(box[0])

    if args["seed"]:
        torch.manual_seed(int(args["seed"]))
    model = get_model(args["network"], in_channels=1, out_channels=1, channel=args["channel"],
                      drop_rate=args["drop_rate"], mode=args["mode"]).to(dev)    # 0.0 still builds the (identity) dropout, like the reference
    if args["train_mode"]:
        model.train_mode = args["train_mode"]
    if model.input_dropout is not None:
        model.input_dropout.seed = int(args["seed"] or 0) * world + rank
    if args["resume"]:                                                   # weights only, from another run's best model (:235-249)
        resume_dir = pathlib.Path(args["resume_dir"] or args["output_dir"]) / method_dir / args["resume"]
        if not (resume_dir / "model" / "best_model.pt.tar").exists():
            raise Exception(f"no checkpoint found at '{args['resume']}'")
        resume(model, resume_dir, dev)

    stego = None if args["covers_only"] else args["stego_method"]
    alpha = None if args["covers_only"] or args["alpha"] is None else float(args["alpha"])
    kw = dict(covers_only=bool(args["covers_only"]), rank=rank, world=world, device=dev, take_num_images=args["take_num_images"])
    per_rank = args["batch_size"] // world
    tr_loader = PairLoader(args["dataset"], args["tr_csv"], stego, alpha, per_rank, shuffle=True, seed=int(args["seed"] or 0), **kw)
    va_loader = PairLoader(args["dataset"], args["va_csv"], stego, alpha, per_rank, shuffle=False, **kw)

    cfg = {k: v for k, v in args.items() if k not in ("mode", "train_mode", "take_num_images")}
    trainer = Trainer(model, loss=args["loss"], lr=args["learning_rate"], out_dir=out_dir, config=cfg, patience=args["patience"])
    best = trainer.fit(tr_loader, va_loader, args["num_epochs"])
    if rank == 0:
        print(f"[train] {out_dir}: best val loss {best:.6f} after {len({e for e, _, _ in trainer.scalars})} epochs")
    return best


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--config", help="a published run's config.json; explicit flags override its keys")
    for key, val in DEFAULTS.items():
        flag = "--" + key
        if isinstance(val, bool):
            ap.add_argument(flag, type=lambda s: s.lower() in ("1", "true", "yes"), default=None)
        elif isinstance(val, list):
            ap.add_argument(flag, type=int, nargs="+", default=None)
        elif isinstance(val, (int, float)) and not isinstance(val, bool):
            ap.add_argument(flag, type=type(val), default=None)
        else:
            ap.add_argument(flag, default=None)
    ns = vars(ap.parse_args(argv))
    args = {}
    cfg_path = ns.pop("config")
    if cfg_path:
        with open(cfg_path) as f:
            args.update({k: v for k, v in json.load(f).items() if k in DEFAULTS})
    args.update({k: v for k, v in ns.items() if v is not None})
    for key in ("num_epochs", "patience", "batch_size", "take_num_images"):
        if args.get(key) is not None:
            args[key] = int(args[key])
    if not args.get("dataset"):
        ap.error("--dataset is required (directory with images*/files.csv, stego*/files.csv and the split CSVs)")
    train(args)


if __name__ == "__main__":
    main()
