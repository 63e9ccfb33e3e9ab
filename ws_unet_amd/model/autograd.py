"""Training path of the UNet: one torch.autograd.Function whose forward and backward are libwsu kernels.

The reference has no hand-written backward -- it relies on autograd through
src/unet/model/unet.py:137-189.  Here the whole network is ONE autograd node: forward saves the
activations, backward walks the layers in reverse calling the K7 kernels (include/wsu.h).  `model.train_mode` picks the path:
  'f16f8p'  (default for planar models) activations AND gradients in the planar three-plane layout (3 bytes per element), the f16f8
            arithmetic in forward, data gradient and weight gradient (_forward_train_pl / _backward_pl); single-plane inputs, no input
            gradient included -- multi-plane inputs fall back to 'bf16x3';
  'bf16x3'  fp32 NHWC tensors; the matrix kernels run the f16f8 arithmetic on them (`train_fwd_mode` / `train_bwd_mode` = 'f16f8x', default)
            or split-bf16; 2-bit pool argmax saved by the forward;
  'f32'     exact fp32 on the matrix cores.
Bias gradients are fp32 sums in every mode.  The gradient w.r.t. the network input (saliency, src/saliency.py:159-174) is produced when
`x.requires_grad`.

Conventions inside backward: `g` is the PRE-activation gradient of the layer being processed; every kernel
that produces the gradient w.r.t. a post-ReLU activation applies that activation's ReLU mask itself
(relu'(0) = 0 as in PyTorch), so `g` can be fed straight to the next weight / data gradient.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from .. import ops
from .unet import ENC, dec_names


def _param_list(model) -> List[torch.nn.Parameter]:
    return [p for _, p in model.named_parameters()]


def _forward_train(model, x: torch.Tensor, m0: int) -> Dict[str, torch.Tensor]:
    """Same dataflow as UNet.forward_features, keeping what backward needs.  The matrix layers of a split-bf16 training run use the
    'f16f8x' arithmetic (exact f16 products + fp8 cross terms on fp32 tensors, ~2^-15 relative: 0.7 of bf16x3's matrix cycles);
    ``model.train_fwd_mode`` / WSU_TRAIN_FWD_MODE='bf16x3' keeps the forward in split-bf16."""
    t: Dict[str, torch.Tensor] = {}
    e11 = model.e11
    cur = t["xe11"] = ops.conv3x3_first(x, e11.weight.detach(), e11.bias.detach(), m0, relu=True)
    m = m0
    if m0 == ops.MODE_BF16X3:
        m = ops.mode_id(getattr(model, "train_fwd_mode", None) or "f16f8x")
    for lvl in range(model.nsteps + 1):
        a, b = ENC[lvl]
        if lvl >= 1:
            la = getattr(model, a)
            cur = t["x" + a] = ops.conv3x3(cur, None, model._packed(a, m, "conv"), la.bias.detach(), la.out_channels, m)
        lb = getattr(model, b)
        if lvl < model.nsteps:
            full, cur, idx = ops.conv3x3(cur, None, model._packed(b, m, "conv"), lb.bias.detach(), lb.out_channels, m,
                                         pool=True, pool_idx=True)
            t["x" + b], t[f"xp{lvl + 1}"], t[f"idx{lvl + 1}"] = full, cur, idx
        else:
            cur = t["x" + b] = ops.conv3x3(cur, None, model._packed(b, m, "conv"), lb.bias.detach(), lb.out_channels, m)
    for depth in range(model.nsteps, 0, -1):
        up, c1, c2 = dec_names(depth)
        lu, l1, l2 = getattr(model, up), getattr(model, c1), getattr(model, c2)
        xu = t["xu" + up[-1]] = ops.convt2x2(cur, model._packed(up, m, "convt"), lu.bias.detach(), lu.out_channels, m)
        skip = t["x" + ENC[depth - 1][1]]
        cur = t["x" + c1] = ops.conv3x3(xu, skip, model._packed(c1, m, "conv"), l1.bias.detach(), l1.out_channels, m)
        cur = t["x" + c2] = ops.conv3x3(cur, None, model._packed(c2, m, "conv"), l2.bias.detach(), l2.out_channels, m)
    t["last"] = cur
    t["out"] = ops.conv1x1_sigmoid(cur, model.outconv.weight.detach(), model.outconv.bias.detach(), m0)
    return t


def _forward_train_pl(model, x: torch.Tensor) -> Dict[str, torch.Tensor]:
    """train_mode 'f16f8p': the inference path's planar kernels (UNet._forward_planar) with every activation the backward needs kept --
    in the planar layout, 3 bytes per element; the last conv stores its output AND the fused head's."""
    W = ops.MODE_F16F8
    t: Dict[str, torch.Tensor] = {}
    e11 = model.e11
    rf = model._range_flag_tensor(x.device)         # OR-ed by every epilogue that stores an activation beyond +-448 (UNet.range_exceeded)
    head = dict(head_w=model.outconv.weight.detach(), head_b=model.outconv.bias.detach())
    # 1-bit ReLU masks (relu_mask planes, include/wsu.h) of every activation whose mask a data gradient applies: written by the producing
    # kernel's epilogue, read by the consumer's loaders by LDS-DMA (1/8 byte per element instead of 2)
    cur, t["m_xe11"] = ops.conv3x3_first_pl(x, e11.weight, e11.bias.detach(), range_flag=rf, want_mask=True)
    t["xe11"] = cur
    for lvl in range(model.nsteps + 1):
        a, b = ENC[lvl]
        if lvl >= 1:
            la = getattr(model, a)
            cur, t["m_x" + a] = ops.conv3x3_pl(cur, None, model._packed(a, W, "conv"), la.bias.detach(), la.out_channels, range_flag=rf, want_mask=True)
            t["x" + a] = cur
        lb = getattr(model, b)
        if lvl < model.nsteps:
            t["x" + b], cur = ops.conv3x3_pl(cur, None, model._packed(b, W, "conv"), lb.bias.detach(), lb.out_channels, pool=True, range_flag=rf)
            t[f"xp{lvl + 1}"] = cur
        elif model.nsteps == 0:
            t["out"], cur = ops.conv3x3_pl(cur, None, model._packed(b, W, "conv"), lb.bias.detach(), lb.out_channels, range_flag=rf, **head)
            t["x" + b] = cur
        else:
            cur = t["x" + b] = ops.conv3x3_pl(cur, None, model._packed(b, W, "conv"), lb.bias.detach(), lb.out_channels, range_flag=rf)
    for depth in range(model.nsteps, 0, -1):
        up, c1, c2 = dec_names(depth)
        lu, l1, l2 = getattr(model, up), getattr(model, c1), getattr(model, c2)
        xu = t["xu" + up[-1]] = ops.convt2x2_pl(cur, model._packed(up, W, "convt"), lu.bias.detach(), lu.out_channels, range_flag=rf)
        cur, t["m_x" + c1] = ops.conv3x3_pl(xu, t["x" + ENC[depth - 1][1]], model._packed(c1, W, "conv"), l1.bias.detach(), l1.out_channels, range_flag=rf, want_mask=True)
        t["x" + c1] = cur
        if depth == 1:
            t["out"], cur = ops.conv3x3_pl(cur, None, model._packed(c2, W, "conv"), l2.bias.detach(), l2.out_channels, range_flag=rf, **head)
            t["x" + c2] = cur
        else:
            cur = t["x" + c2] = ops.conv3x3_pl(cur, None, model._packed(c2, W, "conv"), l2.bias.detach(), l2.out_channels, range_flag=rf)
    t["last"] = cur
    return t


def _backward_pl(model, t: Dict[str, torch.Tensor], x: torch.Tensor, dout: torch.Tensor, want_dx: bool = False) -> Dict[str, torch.Tensor]:
    """Backward of train_mode 'f16f8p': every gradient tensor planar (f16 + e4m3 residual, pre-scaled by a power of two), data gradients through
    the persistent LDS-DMA conv kernel, weight gradients from planar operands; the same layer walk as the fp32-storage path below."""
    W = ops.MODE_F16F8
    products = getattr(model, "train_products", "f16f8")      # which terms the backward matrix kernels multiply (wsu.h WSU_PRODUCTS_*)
    grads: Dict[str, torch.Tensor] = {}
    scale = ops.pow2_grad_scale(dout)
    dout = ops.scale_by(dout, scale[0:1])

    def conv_bwd(name, g, x1, x2, mask1, need_dx=True, mask1_bits=None):
        layer = getattr(model, name)
        grads[name + ".weight"], grads[name + ".bias"] = ops.conv3x3_pl_bwd_weight(g, x1, x2, products=products)
        if not need_dx:
            return None, None
        return ops.conv3x3_pl_bwd_data(g, model._packed(name, W, "dgrad"), model._packed(name, W, "ring"), layer.in_channels,
                                       x1.shape[1] * 16, mask1, None, mask1_bits=mask1_bits, products=products)

    g, grads["outconv.weight"], grads["outconv.bias"] = ops.conv1x1_sigmoid_pl_bwd(t["last"], model.outconv.weight, t["out"], dout, products=products)
    skip_g: Dict[int, torch.Tensor] = {}
    for depth in range(1, model.nsteps + 1):
        up, c1, c2 = dec_names(depth)
        xc1, xu, skip = t["x" + c1], t["xu" + up[-1]], t["x" + ENC[depth - 1][1]]
        g, _ = conv_bwd(c2, g, xc1, None, xc1, mask1_bits=t.get("m_x" + c1))
        dxu, skip_g[depth] = conv_bwd(c1, g, xu, skip, None)              # neither half is masked here: the upconv output has no ReLU, the skip's
        below = t["x" + (dec_names(depth + 1)[2] if depth < model.nsteps else ENC[model.nsteps][1])]    # mask meets the pool routing below
        lu = getattr(model, up)
        grads[up + ".weight"], grads[up + ".bias"] = ops.convt2x2_pl_bwd_weight(below, dxu, products=products)
        g = ops.convt2x2_pl_bwd_data(dxu, model._packed(up, W, "convt_dgrad_pl"), lu.in_channels, below, products=products)
    for lvl in range(model.nsteps, -1, -1):
        a, b = ENC[lvl]
        if lvl < model.nsteps:
            g = ops.maxpool2x2_pl_bwd(skip_g[lvl + 1], g, t["x" + b], products=products)
        xa = t["x" + a]
        g, _ = conv_bwd(b, g, xa, None, xa, mask1_bits=t.get("m_x" + a))
        if lvl == 0:
            grads[a + ".weight"], grads[a + ".bias"] = ops.conv3x3_first_pl_bwd_weight(g, x, products=products)
            if want_dx:                                                   # saliency (src/saliency.py:159-174): the input gradient, in the planar arithmetic too
                grads["__dx__"] = ops.conv3x3_first_pl_bwd_data(g, getattr(model, a).weight, products=products)
        else:
            g, _ = conv_bwd(a, g, t[f"xp{lvl}"], None, None)
    ops.scale_many_(list(grads.values()), scale[1:2])
    return grads


def planar_train_ok(model, x: torch.Tensor) -> bool:
    """The planar training path covers what the planar inference path covers, for single-plane inputs (the input gradient included since
    round 4: wsu_conv3x3_first_pl_bwd_data)."""
    return model._planar_ok() and model.e11.in_channels == 1 and model.outconv.in_channels == 64


def planar_range_fallback(model) -> bool:
    """True (after switching the model to train_mode 'bf16x3', loudly) if a planar forward stored activations beyond +-448 since the last look:
    there the e4m3 residual saturates and the planar format keeps only f16 accuracy; fp32 storage has fp32's range."""
    # collective: in a data-parallel job the ranks must not train in different arithmetics (one MAX all-reduce of one word, first forward only)
    from .. import parallel
    rf = getattr(model, "_range_flag", None)
    if rf is None or not parallel.any_rank_flag(rf):
        return False
    rf.zero_()
    import logging
    logging.warning("ws_unet_amd.UNet: activations beyond +-448 while training in train_mode 'f16f8p' (the planar format's e4m3 residual "
                    "saturates there); switching this model to train_mode 'bf16x3' (fp32 storage)")
    model.train_mode = "bf16x3"
    return True


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        tm = getattr(model, "train_mode", "f32")
        if tm == "f16f8p":
            if planar_train_ok(model, x):
                t = _forward_train_pl(model, x)
                ok = True
                if not getattr(model, "_range_checked_train", False):
                    # first planar training forward of this model: one synchronising look at the range flag, as the inference path does.  The
                    # trainer looks again after every epoch (Trainer._run_epoch): weights move.
                    model._range_checked_train = True
                    ok = not planar_range_fallback(model)
                if ok:
                    ctx.model, ctx.t, ctx.x, ctx.m = model, t, x, ops.MODE_F16F8P
                    return t["out"]
                t = None
            tm = model.train_mode if model.train_mode != "f16f8p" else "bf16x3"   # input gradients (saliency), odd shapes, range fallback: fp32 storage
        m = ops.mode_id(tm)
        if m == ops.MODE_BF16:
            raise ValueError("train_mode must be 'f32' or 'bf16x3' (activations are kept in fp32 for the backward pass)")
        t = _forward_train(model, x, m)
        ctx.model, ctx.t, ctx.x, ctx.m = model, t, x, m
        return t["out"]

    @staticmethod
    def backward(ctx, dout):
        model, t, x, m = ctx.model, ctx.t, ctx.x, ctx.m
        grads: Dict[str, torch.Tensor] = {}
        dx = None
        dout = dout.contiguous().float()
        if m == ops.MODE_F16F8P:
            grads = _backward_pl(model, t, x, dout, want_dx=ctx.needs_input_grad[1])
            ctx.t = None
            return (None, grads.pop("__dx__", None)) + tuple(grads[name] if p.requires_grad else None for name, p in model.named_parameters())
        # data- and weight-gradient GEMMs of a split-bf16 run: f16f8 arithmetic on the fp32 tensors (model.train_bwd_mode).  Gradients of
        # a mean-reduced loss sit far below f16's normal range, so the whole backward chain runs on gradients scaled by a power of two
        # chosen from |dL/dout| (every kernel on the way is linear in the gradient; ReLU masks and pool routing ignore the scale) and all
        # parameter / input gradients are scaled back at the end -- exact, and computed on the device by libwsu (wsu_pow2_grad_scale,
        # wsu_scale_f32, wsu_scale_multi_tensor: no host synchronisation, no ATen arithmetic).
        mb, scale = m, None
        if m == ops.MODE_BF16X3 and (getattr(model, "train_bwd_mode", None) or "f16f8x") == "f16f8x":
            mb = ops.MODE_F16F8X
            # max |dout| * scale in (2, 4]: 2^14 of headroom below f16's largest value for gradients that grow on the way down, while values
            # 2^-27 of that maximum still keep an absolute error below theirs (f16 subnormal spacing 2^-24 + the e4m3 residual)
            scale = ops.pow2_grad_scale(dout)                      # device {scale, 1 / scale}: one reduction + one thread, no ATen chain
            dout = ops.scale_by(dout, scale[0:1])

        def conv_bwd(name, g, x1, x2, mask1, mask2, need_dx=True):
            layer = getattr(model, name)
            grads[name + ".weight"], grads[name + ".bias"] = ops.conv3x3_bwd_weight(g, x1, x2, mode=mb)
            if not need_dx:
                return None, None
            csplit = x1.shape[3]
            return ops.conv3x3_bwd_data(g, model._packed(name, mb, "dgrad"), layer.weight, csplit, mask1, mask2, mb)

        g, grads["outconv.weight"], grads["outconv.bias"] = ops.conv1x1_sigmoid_bwd(t["last"], model.outconv.weight, t["out"], dout)
        skip_g: Dict[int, torch.Tensor] = {}
        for depth in range(1, model.nsteps + 1):
            up, c1, c2 = dec_names(depth)
            xc1, xu, skip = t["x" + c1], t["xu" + up[-1]], t["x" + ENC[depth - 1][1]]
            g, _ = conv_bwd(c2, g, xc1, None, xc1, None)                        # -> pre-activation grad of c1
            dxu, skip_g[depth] = conv_bwd(c1, g, xu, skip, None, skip)           # upconv output has no ReLU; skip is masked
            below = t["x" + (dec_names(depth + 1)[2] if depth < model.nsteps else ENC[model.nsteps][1])]
            lu = getattr(model, up)
            grads[up + ".weight"], grads[up + ".bias"] = ops.convt2x2_bwd_weight(below, dxu, mode=mb)
            g = ops.convt2x2_bwd_data(dxu, model._packed(up, m, "convt_dgrad"), lu.in_channels, below, m)
        for lvl in range(model.nsteps, -1, -1):
            a, b = ENC[lvl]
            if lvl < model.nsteps:
                # g is the gradient w.r.t. the pooled tensor xp{lvl+1}; route it onto the argmax and add the skip path
                g = ops.maxpool2x2_bwd(skip_g[lvl + 1], g, t[f"idx{lvl + 1}"], t[f"xp{lvl + 1}"])
            xa = t["x" + a]
            g, _ = conv_bwd(b, g, xa, None, xa, None)                            # -> pre-activation grad of conv a
            if lvl == 0:
                grads[a + ".weight"], grads[a + ".bias"] = ops.conv3x3_first_bwd_weight(g, x)
                if ctx.needs_input_grad[1]:                                      # saliency: src/saliency.py:159-174
                    dx = ops.conv3x3_first_bwd_data(g, getattr(model, a).weight)
            else:
                g, _ = conv_bwd(a, g, t[f"xp{lvl}"], None, None, None)           # pooled tensor: no ReLU of its own
        ctx.t = None
        if scale is not None:
            ops.scale_many_(list(grads.values()) + [dx], scale[1:2])     # every parameter / input gradient back to its true scale: ONE launch
        out = [None, dx if ctx.needs_input_grad[1] else None]
        for name, p in model.named_parameters():
            out.append(grads[name] if p.requires_grad else None)
        return tuple(out)


def unet_apply(model, x: torch.Tensor) -> torch.Tensor:
    return _UNetFn.apply(model, x, *_param_list(model))
