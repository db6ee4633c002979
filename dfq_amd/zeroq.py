"""ZeroQ data distillation (ZeroQ/distill_data.py) with the BatchNorm-statistics loss on the HIP engine.

``bn_stat_losses(x, bn_mean, bn_std)`` is the pair of losses the reference accumulates per BN layer
(distill_data.py:170-190: spatial mean / unbiased std of the BN input per sample and channel against the BN's
running statistics, ``own_loss`` :40-45) as ONE autograd node: one read of the activation forward, one read +
one write backward, instead of ~10 eager passes.  ``getDistilData`` is the reference's optimisation loop
(:75-227) around it; the convolutions' forward / backward stay PyTorch's (MIOpen).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.optim as optim

from . import _ffi


class _BNStatLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bn_mean, bn_std, eps, denom):
        lib = _ffi.lib()
        stage = _ffi.Stage()
        xx = stage.bind(x)
        n, c = xx.shape[0], xx.shape[1]
        hw = xx[0, 0].numel()
        m, s = stage.bind(bn_mean).reshape(-1), stage.bind(bn_std).reshape(-1)
        assert m.numel() == c and s.numel() == c, 'BN statistics must have one entry per channel'
        row_mean, row_std = stage.new((n * c,)), stage.new((n * c,))
        loss2 = stage.new((2,))
        scratch = stage.new((int(lib.dfq_bn_stat_loss_scratch_bytes(n * c)) // 8 + 1,), dtype=torch.float64)
        _ffi.check(lib.dfq_bn_stat_loss_forward(_ffi.ptr(xx), n * c, hw, c, _ffi.ptr(m), _ffi.ptr(s), float(eps), float(denom),
                                                _ffi.ptr(row_mean), _ffi.ptr(row_std), _ffi.ptr(loss2), _ffi.ptr(scratch),
                                                _ffi.stream_arg()))
        ctx.save_for_backward(xx, m, s, row_mean, row_std)
        ctx.eps = float(eps)
        ctx.denom = float(denom)
        ctx.shape = x.shape
        ctx.src_device = x.device
        out = stage.out_like(x, loss2)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_mean, g_std):
        lib = _ffi.lib()
        xx, m, s, row_mean, row_std = ctx.saved_tensors
        n, c = xx.shape[0], xx.shape[1]
        hw = xx[0, 0].numel()
        grad = torch.empty_like(xx)
        # the upstream gradients stay on the device (no .item() inside backward: ~50 BN layers per iteration would each
        # stall the stream); an output that took no part in the loss arrives as None = 0
        zero = torch.zeros((), dtype=torch.float32, device=xx.device)
        pair = torch.stack([(zero if g is None else g.detach().to(device=xx.device, dtype=torch.float32).reshape(()))
                            for g in (g_mean, g_std)]).contiguous()
        _ffi.check(lib.dfq_bn_stat_loss_backward_dev(_ffi.ptr(xx), n * c, hw, c, _ffi.ptr(m), _ffi.ptr(s), ctx.eps, ctx.denom,
                                                     _ffi.ptr(row_mean), _ffi.ptr(row_std), _ffi.ptr(pair), _ffi.ptr(grad), 0,
                                                     _ffi.stream_arg()))
        return grad.to(ctx.src_device).reshape(ctx.shape), None, None, None, None


def bn_stat_losses(x, bn_mean, bn_std, eps=1e-6, denom=None):
    """(mean_loss, std_loss) of one BN input ``x`` [N, C, H, W] (distill_data.py:172-190), differentiable in x.
    ``denom`` is own_loss's ``A.size(0)``: the channel count for the BN terms (default), the batch size for the
    input-batch term (:192-196, where the per-sample statistics come first)."""
    return _BNStatLoss.apply(x, bn_mean, bn_std, eps, x.shape[1] if denom is None else denom)


class _InputHook:
    def __init__(self):
        self.inputs = None

    def hook(self, module, input, output):
        self.inputs = input

    def clear(self):
        self.inputs = None


def getDistilData(teacher_model, shape, num_batch=1, bn_merged=False, value_range=(-10, 10), max_value=3.,
                  early_break_factor=1., iterations=1000, generator=None, init=None, loss_log=None):
    """The reference's distillation loop (distill_data.py:75-227).  ``shape`` = (batch, 3, H, W) replaces its
    dataset switch; the start is uniform noise in [-max_value, max_value] like its ``UniformDataset``, or the
    tensors of ``init`` (one per batch: what the reference's data loader would have produced).  ``loss_log`` (a list)
    receives the total loss of every iteration (the value the reference hands to its LR scheduler)."""
    eps = 1e-6
    dev = next(teacher_model.parameters()).device
    teacher_model = teacher_model.eval()
    hooks, handles, bn_stats = [], [], []
    for m in teacher_model.modules():
        if isinstance(m, nn.BatchNorm2d):
            h = _InputHook()
            hooks.append(h)
            handles.append(m.register_forward_hook(h.hook))
            if not bn_merged:
                bn_stats.append((m.running_mean.detach().clone().flatten(), torch.sqrt(m.running_var + eps).detach().clone().flatten()))
            else:
                bn_stats.append((m.fake_bias.detach().clone().flatten(), m.fake_weight.detach().clone().flatten()))
    layers = len(hooks)
    refined = []
    for b in range(num_batch):
        if init is not None:
            data = init[b].detach().clone().to(dev)
        else:
            data = ((torch.rand(*shape, generator=generator) * 2 - 1) * max_value).to(dev)
        data.requires_grad = True
        optimizer = optim.Adam([data], lr=0.1)
        scheduler = optim.lr_scheduler.ReduceLROnPlateau(optimizer, min_lr=1e-7, patience=100)
        in_mean = torch.zeros(3, device=dev)
        in_std = torch.ones(3, device=dev)
        for it in range(iterations):
            teacher_model.zero_grad()
            optimizer.zero_grad()
            for h in hooks:
                h.clear()
            teacher_model(data.clamp(value_range[0], value_range[1]))
            mean_loss, std_loss = 0, 0
            for (bn_mean, bn_std), h in zip(bn_stats, hooks):
                ml, sl = bn_stat_losses(h.inputs[0], bn_mean, bn_std, eps)
                mean_loss = mean_loss + ml
                std_loss = std_loss + sl
            ml, sl = bn_stat_losses(data, in_mean, in_std, 0.0, denom=data.shape[0])     # :192-196 (no eps, / N)
            total = mean_loss + ml + std_loss + sl
            total.backward()
            optimizer.step()
            total_value = total.item()
            if loss_log is not None:
                loss_log.append(total_value)
            scheduler.step(total_value)
            if total_value <= (layers + 1) * early_break_factor:
                break
        refined.append(data.detach().clone().clamp(value_range[0], value_range[1]))
    for hd in handles:
        hd.remove()
    return refined
