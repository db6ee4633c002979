"""Row f2: the BatchNorm-statistics loss of ZeroQ's data distillation (ZeroQ/distill_data.py:40-45, :172-196).

Pinned by the reference itself: tests/golden/zeroq_*.npz hold k iterations of the UNMODIFIED `getDistilData` (stub-
imported, oracle/make_golden_zeroq.py) on a small teacher network -- start batch, per-iteration losses, refined batch --
and `dfq_amd.zeroq.getDistilData` must reproduce them (1e-4).  Kernel-level three-way check besides: the HIP kernels
(CPU emulation / MI355X) against the float64 oracle and against the torch expressions of distill_data.py:172-190."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from dfq_amd import zeroq
from oracle import dfq_oracle as orc
from tests.common import GOLD, build_distill_net, npy


def _reference_losses(tmp_output, bn_mean, bn_std, eps=1e-6):
    """distill_data.py:172-190 verbatim in meaning (own_loss :40-45)."""
    own_loss = lambda A, B: (A - B).norm() ** 2 / A.size(0)
    tmp_mean = torch.mean(tmp_output.view(tmp_output.size(0), tmp_output.size(1), -1), dim=2)
    tmp_std = torch.std(tmp_output.view(tmp_output.size(0), tmp_output.size(1), -1) + eps, dim=2)
    return own_loss(bn_mean, tmp_mean), own_loss(bn_std, tmp_std)


@pytest.mark.parametrize('shape', [(2, 3, 7, 7), (4, 16, 14, 14), (1, 5, 40, 40), (3, 2, 1, 2)])
def test_losses_and_gradients(engine, shape):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g) * 1.5 + 0.3
    bn_mean = torch.randn(shape[1], generator=g)
    bn_std = torch.rand(shape[1], generator=g) + 0.5
    # reference arithmetic with autograd (CPU, float32)
    xr = x.clone().requires_grad_(True)
    ml_r, sl_r = _reference_losses(xr, bn_mean, bn_std)
    (ml_r * 0.7 + sl_r * 1.3).backward()
    # engine
    xe = engine.to(x.clone()).requires_grad_(True)
    ml, sl = zeroq.bn_stat_losses(xe, engine.to(bn_mean), engine.to(bn_std))
    (ml * 0.7 + sl * 1.3).backward()
    # oracle (float64)
    ml_o, sl_o, gm_o, gs_o = orc.bn_stat_losses(x.numpy(), bn_mean.numpy(), bn_std.numpy())
    for got, ref, ora in ((float(ml.detach()), float(ml_r.detach()), ml_o), (float(sl.detach()), float(sl_r.detach()), sl_o)):
        assert abs(got - ora) <= 1e-5 * max(1.0, abs(ora))
        assert abs(got - ref) <= 1e-4 * max(1.0, abs(ref))          # the reference's float32 reductions
    grad_o = 0.7 * gm_o + 1.3 * gs_o
    np.testing.assert_allclose(npy(xe.grad), grad_o, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(npy(xe.grad), xr.grad.numpy(), rtol=2e-3, atol=1e-5)


def test_input_batch_term_divides_by_batch(engine):
    """distill_data.py:192-196: own_loss(tmp_mean [N, 3], input_mean) divides by N, and no eps is added."""
    g = torch.Generator().manual_seed(5)
    data = torch.randn(4, 3, 16, 16, generator=g)
    tmp_mean = torch.mean(data.view(4, 3, -1), dim=2)
    tmp_std = torch.std(data.view(4, 3, -1), dim=2)
    own_loss = lambda A, B: (A - B).norm() ** 2 / A.size(0)
    want = (float(own_loss(tmp_mean, torch.zeros(1, 3))), float(own_loss(tmp_std, torch.ones(1, 3))))
    ml, sl = zeroq.bn_stat_losses(engine.to(data), engine.to(torch.zeros(3)), engine.to(torch.ones(3)), 0.0, denom=4)
    assert abs(float(ml) - want[0]) <= 1e-4 * max(1, want[0]) and abs(float(sl) - want[1]) <= 1e-4 * max(1, want[1])


def test_single_pixel_maps_follow_the_reference_branch(engine):
    """H*W == 1 (distill_data.py:181-182): the mean term per value, the std term over the [N, C] block viewed as [C, N]."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 4, 1, 1, generator=g)
    bn_mean, bn_std = torch.randn(4, generator=g), torch.rand(4, generator=g) + 0.5
    own_loss = lambda A, B: (A - B).norm() ** 2 / A.size(0)
    xr = x.clone().requires_grad_(True)
    tmp_mean = torch.mean(xr.view(5, 4, -1), dim=2)
    tmp_std = torch.std(xr.view(4, -1) + 1e-6, dim=1)
    ml_r, sl_r = own_loss(bn_mean, tmp_mean), own_loss(bn_std, tmp_std)
    (ml_r * 0.7 + sl_r * 1.3).backward()
    xe = engine.to(x.clone()).requires_grad_(True)
    ml, sl = zeroq.bn_stat_losses(xe, engine.to(bn_mean), engine.to(bn_std))
    (ml * 0.7 + sl * 1.3).backward()
    assert abs(float(ml.detach()) - float(ml_r.detach())) <= 1e-5 * max(1.0, float(ml_r.detach()))
    assert abs(float(sl.detach()) - float(sl_r.detach())) <= 1e-5 * max(1.0, float(sl_r.detach()))
    np.testing.assert_allclose(npy(xe.grad), xr.grad.numpy(), rtol=1e-4, atol=1e-6)
    from dfq_amd import _ffi
    with pytest.raises(_ffi.DfqError):          # one sample: the reference's std over the batch axis is NaN; refused here
        zeroq.bn_stat_losses(engine.to(torch.randn(1, 4, 1, 1)), engine.to(torch.zeros(4)), engine.to(torch.ones(4)))


def test_unused_output_gets_zero_gradient(engine):
    """Only one of the two losses takes part in the objective: autograd hands None for the other."""
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 6, 6, generator=g)
    bn_mean, bn_std = torch.randn(3, generator=g), torch.rand(3, generator=g) + 0.5
    xe = engine.to(x.clone()).requires_grad_(True)
    ml, _ = zeroq.bn_stat_losses(xe, engine.to(bn_mean), engine.to(bn_std))
    ml.backward()
    xr = x.clone().requires_grad_(True)
    _reference_losses(xr, bn_mean, bn_std)[0].backward()
    np.testing.assert_allclose(npy(xe.grad), xr.grad.numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('case', ['zeroq_s0', 'zeroq_s1_px'])
def test_distillation_against_reference(engine, case):
    """k iterations of the reference's getDistilData vs dfq_amd.zeroq.getDistilData from the same start batch."""
    import os
    gold = np.load(os.path.join(GOLD, case + '.npz'))
    seed, k, px = (int(v) for v in gold['cfg'])
    net = build_distill_net(torch.Generator().manual_seed(seed), bool(px))
    net.load_state_dict({n[len('param.'):]: torch.from_numpy(gold[n]) for n in gold.files if n.startswith('param.')})
    net.to(engine.device)
    start = torch.from_numpy(gold['start'])
    losses = []
    out = zeroq.getDistilData(net, tuple(start.shape), num_batch=1, iterations=k, init=[start], early_break_factor=0.0,
                              loss_log=losses)
    assert len(losses) == k
    np.testing.assert_allclose(np.array(losses), gold['losses'], rtol=1e-4)
    # Adam's first steps are +-lr regardless of the gradient's size, so the batches stay together to float32 noise
    np.testing.assert_allclose(npy(out[0]), gold['refined'], rtol=0, atol=1e-4)


def test_distillation_loop_reduces_the_loss(engine):
    """getDistilData (distill_data.py:75-227) on a small conv net: the statistics loss of the distilled batch is far
    below that of the noise it started from."""
    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1, stride=2),
                        nn.BatchNorm2d(8), nn.ReLU()).to(engine.device).eval()
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.5)
            m.running_var.uniform_(0.5, 1.5)
    gen = torch.Generator().manual_seed(1)

    def total_loss(batch):
        hooks = []
        acc = [0.0]
        def mk(m):
            def hook(mod, inp, out):
                a, b = zeroq.bn_stat_losses(inp[0], mod.running_mean, torch.sqrt(mod.running_var + 1e-6))
                acc[0] += float(a) + float(b)
            return m.register_forward_hook(hook)
        hs = [mk(m) for m in net.modules() if isinstance(m, nn.BatchNorm2d)]
        with torch.no_grad():
            net(batch)
        for h in hs:
            h.remove()
        return acc[0]
    start = ((torch.rand(4, 3, 16, 16, generator=torch.Generator().manual_seed(1)) * 2 - 1) * 3.).to(engine.device)
    out = zeroq.getDistilData(net, (4, 3, 16, 16), num_batch=1, iterations=60, generator=gen, early_break_factor=0.0)
    assert len(out) == 1 and out[0].shape == (4, 3, 16, 16)
    assert total_loss(out[0]) < 0.7 * total_loss(start)          # (the default early break stops at loss <= #BN + 1)
