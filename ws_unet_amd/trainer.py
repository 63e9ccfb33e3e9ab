"""Reconstructed UNet train step and epoch loop (the reference publishes none: SURVEY.md F2 / 3.4).

Pattern taken from the detector's loop (src/detector/train.py:55-95 train_one_epoch, :228 AdamW(params, lr),
:281-304 checkpoint / best copy / patience) with the UNet-specific pieces: criterion(outputs, (covers, alphas),
inputs) from src/_defs/losses.py, meters MAEMeter / WSMeter (src/_defs/metrics.py:64-142), scalar tags
train|val/{loss, mae, ws}, run naming src/_defs/defs.py:47-74, config.json schema of models/unet/*/.

One step = zero_grad -> forward -> loss -> backward -> [sum all-reduce of the flat gradient bucket] -> AdamW.
Every stage is a libwsu kernel; parameters, gradients and AdamW moments live in three flat fp32 buffers
(parameters are views into the first), so the optimiser is ONE kernel launch and DP needs ONE collective.
"""
from __future__ import annotations

import json
import shutil
import time
from pathlib import Path
from typing import Dict, Optional

import numpy as np
import torch

from . import losses, metrics, ops, parallel


def create_run_name(args: Dict) -> str:
    """Run-directory suffix, same composition rules as src/_defs/defs.py:47-74."""
    parts = [str(args["network"])]
    if args.get("no_stem_stride"):
        parts[0] += "-nostride"
    name = parts[0] + "-"
    if args.get("alpha"):
        name += f"alpha_{args['alpha']}_"
    name += "grayscale_" if args.get("grayscale") else "color_" + "".join(map(str, args.get("channel", [])))
    if args.get("demosaic"):
        name += "_".join(args["demosaic"]) + "_"
    if args.get("demosaic_oracle"):
        name += "oracle_"
    if args.get("loss"):
        name += args["loss"] + "_"
        if args["loss"] == "l1ws":
            name += f"{args['loss_lambda']:.02f}_"
    if args.get("learning_rate"):
        name += f"lr_{args['learning_rate']}_"
    if args.get("drop_rate"):
        name += f"dr_{args['drop_rate']}"
    return name


class FlatAdamW:
    """AdamW over one flat parameter buffer (defaults of torch.optim.AdamW: betas (.9,.999), eps 1e-8, wd 1e-2)."""

    def __init__(self, model: torch.nn.Module, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        self.model, self.lr, self.betas, self.eps, self.weight_decay = model, lr, betas, eps, weight_decay
        params = [p for p in model.parameters() if p.requires_grad]
        dev = params[0].device
        n = sum(p.numel() for p in params)
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            k = p.numel()
            self.flat_param[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + k].view(p.shape)          # parameters become views of the flat buffer
            p.grad = self.flat_grad[off:off + k].view(p.shape)           # autograd accumulates into the flat bucket
            off += k
        self.params = params
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.step_count = 0
        self._table = ops.AdamWTable([self.flat_param], [self.flat_grad], [self.exp_avg], [self.exp_avg_sq])
        if hasattr(model, "invalidate_packed"):
            model.invalidate_packed()

    def zero_grad(self):
        self.flat_grad.zero_()
        off = 0
        for p in self.params:                                           # re-attach views if something replaced .grad
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + off * 4:
                p.grad = self.flat_grad[off:off + k].view(p.shape)
            off += k

    def step(self, grad_scale: float = 1.0, skip_flag: Optional[torch.Tensor] = None):
        self.step_count += 1
        self._table.step(self.step_count, self.lr, self.betas, self.eps, self.weight_decay, grad_scale, skip_flag)
        if hasattr(self.model, "invalidate_packed"):
            self.model.invalidate_packed(recheck_range=False)            # flat update bypasses tensor version counters

    def forget_skipped(self, n: int) -> None:
        """`n` of the counted steps were suppressed on the device (skip_flag): the bias corrections follow the APPLIED updates, as
        torch.optim.AdamW's would on the same sequence.  Called by the trainer where it reads the device-side counter (once per epoch),
        so inside an epoch bc1 / bc2 run at most the epoch's skipped steps ahead."""
        self.step_count = max(0, self.step_count - int(n))

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


class Trainer:
    def __init__(self, model: torch.nn.Module, loss: str = "l1ws", lr: float = 1e-4, out_dir: Optional[Path] = None,
                 config: Optional[Dict] = None, patience: int = 10):
        self.model = model
        self.criterion = losses.get_loss(loss)
        self.opt = FlatAdamW(model, lr)
        self.rank, self.world = parallel.world_info()
        parallel.broadcast_parameters(model, flat=self.opt.flat_param)   # identical replicas before the first step: ONE 7.45 MB broadcast
        self.out_dir = Path(out_dir) if out_dir else None
        self.config = dict(config or {})
        self.patience0 = self.patience = patience
        self.best_val_loss = np.inf
        self.scalars = []                                                # (epoch, tag, value) rows, tags as in the tfevents
        self.epochs_run = 0
        self._nonfinite = torch.zeros(2, dtype=torch.int32, device=self.opt.flat_param.device)   # [this step, skipped steps so far]
        self._last_l1 = None
        self._skipped_seen = 0
        if self.out_dir and self.rank == 0:
            (self.out_dir / "model").mkdir(parents=True, exist_ok=True)
            (self.out_dir / "log").mkdir(parents=True, exist_ok=True)
            with open(self.out_dir / "config.json", "w") as f:
                json.dump(self.config, f, indent=4, sort_keys=True)

    # ---- one optimisation step on this rank's shard of the global batch --------------------------------
    def train_step(self, inputs: torch.Tensor, covers: torch.Tensor, alphas: torch.Tensor):
        self.opt.zero_grad()
        outputs = self.model(inputs)
        loss = self.criterion(outputs, (covers, alphas), inputs)
        self._last_l1 = self.criterion.last_parts[0]
        loss.backward()
        scale = parallel.allreduce_flat_(self.opt.flat_grad)             # C1: one 7.45 MB fp32 bucket over xGMI
        # finite guard behind the all-reduce (so every rank takes the same decision): an overflowed f16f8x gradient chain or an inf / NaN
        # input must not reach the AdamW moments -- the step is skipped on the device, counted, and reported once per epoch
        ops.nonfinite_flag(self.opt.flat_grad, self._nonfinite)
        self.opt.step(grad_scale=scale, skip_flag=self._nonfinite)
        return loss.detach(), outputs.detach()

    @torch.no_grad()
    def eval_step(self, inputs, covers, alphas):
        outputs = self.model(inputs)
        loss, _, parts, _ = ops.l1ws_loss_fwd_bwd(outputs, covers.contiguous(), inputs.contiguous(),
                                                   torch.as_tensor(alphas, dtype=torch.float32, device=outputs.device),
                                                   self.criterion.use_l1, self.criterion.use_ws)
        self._last_l1 = parts[0]
        return loss, outputs

    def step_meters(self, inputs, covers, outputs, alphas, loss) -> torch.Tensor:
        """[loss, mae, ws] of one batch as a DEVICE fp64 tensor, with the batch values the reference's meters take
        (src/_defs/metrics.py: MAEMeter(multiplier=1) = mean |cover - output|, WSMeter = mean |clip(beta_hat, 0) - alpha/2| on
        the interior crop).  Nothing but three scalars per batch ever leaves the device."""
        if self._last_l1 is not None and self.criterion.use_l1 != 2:    # the fused loss kernel already reduced mean |cover - out|
            mae = self._last_l1.double()
        else:
            mae = (covers - outputs).abs().mean(dtype=torch.float64)
        beta = ops.ws_meter_beta(inputs.contiguous(), outputs.contiguous())
        ws = (beta.clamp_min(0) - alphas.double() / 2.).abs().mean()
        return torch.stack([loss.detach().double().reshape(()), mae.reshape(()), ws])

    def _run_epoch(self, loader, train: bool, epoch: int):
        dev = next(self.model.parameters()).device
        acc = torch.zeros(3, dtype=torch.float64, device=dev)           # sum loss*n, sum mae, sum ws
        nimg, nbatch = 0, 0
        for inputs, (covers, alphas) in loader:
            inputs, covers = inputs.to(dev, non_blocking=True), covers.to(dev, non_blocking=True)
            alphas = torch.as_tensor(alphas, dtype=torch.float32, device=dev)
            loss, outputs = (self.train_step if train else self.eval_step)(inputs.clone(), covers, alphas)
            m = self.step_meters(inputs, covers, outputs, alphas, loss)
            acc += m * torch.tensor([float(inputs.shape[0]), 1.0, 1.0], dtype=torch.float64, device=dev)
            nimg += inputs.shape[0]
            nbatch += 1
        # the planar format's range flag (activations beyond +-448 stored during this epoch's planar forwards, UNet.range_exceeded) rides on the
        # same reduction: every rank sees the same count and takes the same decision
        rf = getattr(self.model, "_range_flag", None)
        flag = rf.to(torch.float64).reshape(1) if (rf is not None and rf.device == dev) else torch.zeros(1, dtype=torch.float64, device=dev)
        sums = torch.cat([acc, torch.tensor([float(nimg), float(nbatch)], dtype=torch.float64, device=dev), flag])
        tot = parallel.reduce_epoch_sums_(sums).cpu().numpy()            # global sums on every rank; the epoch's only device -> host copy
        nimg, nbatch = tot[3], tot[4]
        if tot[5] > 0:
            import logging
            if rf is not None:
                rf.zero_()
            # every rank sees the same reduced flag and takes the same decision; validation / inference forwards of this model leave the
            # planar format too (model.mode), and the switch is always logged
            was = (getattr(self.model, "train_mode", None), getattr(self.model, "mode", None))
            if was[0] == "f16f8p":
                self.model.train_mode = "bf16x3"
            if was[1] in ("f16f8p", "f16f8q", "f16f4p"):
                self.model.mode = "bf16x3s"
            logging.warning("ws_unet_amd.Trainer: activations beyond +-448 during %s epoch %d (the planar format's e4m3 residual saturates "
                            "there): train_mode %s -> %s, mode %s -> %s", "training" if train else "validation", epoch, was[0],
                            getattr(self.model, "train_mode", None), was[1], getattr(self.model, "mode", None))
        avg = {"loss": tot[0] / max(nimg, 1), "mae": tot[1] / max(nbatch, 1), "ws": tot[2] / max(nbatch, 1)}
        prefix = "train/" if train else "val/"
        for name in ("loss", "mae", "ws"):
            self.scalars.append((epoch, prefix + name, float(avg[name])))
        if train:
            skipped = self.skipped_steps()
            new_skips = skipped - self._skipped_seen
            if new_skips > 0:
                import logging
                logging.warning("ws_unet_amd.Trainer: %d optimiser step(s) skipped in epoch %d (inf / NaN in the gradient bucket; %d so far) -- "
                                "a run that keeps skipping is stalled", new_skips, epoch, skipped)
                self.opt.forget_skipped(new_skips)
                self._skipped_seen = skipped
            self.scalars.append((epoch, "train/skipped_steps", float(skipped)))
        return float(avg["loss"])

    def skipped_steps(self) -> int:
        """Optimiser steps skipped so far because the gradient bucket held an inf / NaN (one device read)."""
        return int(self._nonfinite[1].item())

    def save_checkpoint(self, epoch: int, val_loss: float):
        """Checkpoint dict keys and files as src/detector/train.py:281-296 writes them."""
        if not self.out_dir or self.rank != 0:
            return
        latest = self.out_dir / "model" / "latest_model.pt.tar"
        torch.save({
            "epoch": epoch,
            "state_dict": {k: v.detach().clone() for k, v in self.model.state_dict().items()},
            "best_val_loss": self.best_val_loss,
            "patience": self.patience,
            # the checkpoint keeps the reference's six keys (src/detector/train.py:281-288); the count of optimiser steps the finite guard
            # suppressed rides inside the optimiser state ("step" counts APPLIED updates, FlatAdamW.forget_skipped)
            "optimizer": {**self.opt.state_dict(), "skipped_steps": self._skipped_seen},
            "scheduler": None,
        }, latest)
        if val_loss < self.best_val_loss:
            shutil.copyfile(latest, self.out_dir / "model" / "best_model.pt.tar")
        with open(self.out_dir / "log" / "scalars.csv", "w") as f:
            f.write("epoch,tag,value\n" + "".join(f"{e},{t},{v}\n" for e, t, v in self.scalars))

    def fit(self, tr_loader, va_loader, num_epochs: int):
        for epoch in range(num_epochs):
            if hasattr(tr_loader, "reshuffle"):
                tr_loader.reshuffle()                                    # detector/train.py:255
            self._run_epoch(tr_loader, True, epoch)
            val_loss = self._run_epoch(va_loader, False, epoch)
            self.save_checkpoint(epoch, val_loss)
            # val_loss is the all-reduced average over EVERY rank's validation shard (_run_epoch), so all ranks take the same
            # branch here and leave the loop together (train.py:298-304)
            self.best_val_loss, self.patience, stop = parallel.early_stop_update(val_loss, self.best_val_loss, self.patience, self.patience0)
            self.epochs_run = epoch + 1
            if stop:
                break
        return self.best_val_loss


def resume(model: torch.nn.Module, run_dir: Path, device) -> int:
    """Load only the weights of another run's best model (src/detector/train.py:235-249)."""
    ckpt = torch.load(Path(run_dir) / "model" / "best_model.pt.tar", map_location=device, weights_only=True)
    model.load_state_dict(ckpt["state_dict"])
    return int(ckpt["epoch"])


def synthetic_pretrain(model: torch.nn.Module, steps: int = 300, batch: int = 8, size: int = 128, seed: int = 5, lr: float = 1e-3,
                       loss: str = "l1ws") -> float:
    """`steps` AdamW steps of this package's own training loop on synthetic cover / LSBR-stego pairs (formula.synthetic_images, every second
    image embedded at alpha = 0.4; the pattern of detector/train.py:55-95 on one fixed batch).  Gives a model "trained-like" weights: the second
    weight set the accuracy gate of the default inference mode is measured on besides the 'he' formula weights (the reference ships no UNet
    checkpoint: .MISSING_LARGE_BLOBS:7-12) -- tests/test_gpu_round4.py, bench.py `mae_vs_cpu_oracle_trained`.  Returns the last loss."""
    from . import formula
    dev = next(model.parameters()).device
    cov = formula.synthetic_images(batch, size, size, seed=seed)
    st = np.stack([formula.lsbr_embed(c, 0.4, seed=i) if i % 2 else c for i, c in enumerate(cov)])
    covers = ops.u8_to_unit(torch.from_numpy(cov).to(dev))[:, None].contiguous()
    inputs = ops.u8_to_unit(torch.from_numpy(st).to(dev))[:, None].contiguous()
    alphas = torch.tensor([0.4 if i % 2 else 0.0 for i in range(batch)], device=dev)
    tr = Trainer(model, loss=loss, lr=lr)
    last = None
    for _ in range(steps):
        last, _ = tr.train_step(inputs, covers, alphas)
    return float(last.item())
