"""Pin row f2 (ZeroQ data distillation, ZeroQ/distill_data.py:75-227) against the UNMODIFIED reference and write
tests/golden/zeroq_*.npz.

Runs only in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_zeroq.py

`ZeroQ/distill_data.py` pulls in, through `from ZeroQ.utils import *` and `from improve_dfq import GradHook`, four
packages that are absent here and that `getDistilData` never touches: pytorchcv (model zoo), torchvision (datasets),
the un-vendored PyTransformer and tensorboardX.  They are injected into sys.modules as empty stubs and the reference
module is imported as it is.  Three module globals of the imported module are replaced for the run (environment, not
source): `getRandomData` returns the seeded start batch of the fixture instead of a DataLoader over torch's global RNG,
`range` caps the hard-coded 1000 iterations (distill_data.py:159) at k, ReduceLROnPlateau.step records the loss
it is given and its constructor swallows the `verbose=` argument this torch no longer has.  The network is a small conv/BN stack whose last BN sits on 1x1 feature maps, so the reference's
H*W == 1 branch (:181-182) is on the path.
"""
from __future__ import annotations

import builtins
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)
sys.dont_write_bytecode = True

import torch                                   # noqa: E402
import torch.nn as nn                          # noqa: E402
import torch.nn.functional as F               # noqa: E402,F401

for name in ('PyTransformer', 'PyTransformer.transformers', 'PyTransformer.transformers.torchTransformer', 'tensorboardX',
             'pytorchcv', 'pytorchcv.models', 'pytorchcv.models.common', 'pytorchcv.models.shufflenetv2',
             'torchvision', 'torchvision.datasets', 'torchvision.transforms'):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules['PyTransformer.transformers.torchTransformer'].TorchTransformer = type('TorchTransformer', (), {})
sys.modules['tensorboardX'].SummaryWriter = type('SummaryWriter', (), {})
sys.modules['pytorchcv.models.common'].ConvBlock = type('ConvBlock', (nn.Module,), {})
sys.modules['pytorchcv.models.shufflenetv2'].ShuffleUnit = type('ShuffleUnit', (nn.Module,), {})
sys.modules['pytorchcv.models.shufflenetv2'].ShuffleInitBlock = type('ShuffleInitBlock', (nn.Module,), {})
sys.modules['torchvision'].datasets = sys.modules['torchvision.datasets']
sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']

from ZeroQ import distill_data as ref_dd       # noqa: E402  (reference, unmodified)

sys.path.insert(2, os.path.join(ROOT, 'tests'))
from common import build_distill_net           # noqa: E402  (the same builder the tests use)

GOLD = os.path.join(ROOT, 'tests', 'golden')


def run_case(seed, k, shape, with_pixel_bn):
    g = torch.Generator().manual_seed(seed)
    net = build_distill_net(g, with_pixel_bn)
    start = ((torch.randint(high=255, size=shape, generator=g).float() - 127.) / 128.) * 3.0   # UniformDataset, data_utils.py:47
    losses = []
    Plateau = torch.optim.lr_scheduler.ReduceLROnPlateau
    orig_init, orig_step = Plateau.__init__, Plateau.step

    def init_without_verbose(self, *a, verbose=False, **kw):      # torch >= 2.7 dropped the `verbose` argument (distill_data.py:162)
        return orig_init(self, *a, **kw)

    def recording_step(self, metrics, *a, **kw):
        losses.append(float(metrics))
        return orig_step(self, metrics, *a, **kw)
    ref_dd.getRandomData = lambda **kw: [start.clone()]
    ref_dd.range = lambda n: builtins.range(k)
    Plateau.__init__, Plateau.step = init_without_verbose, recording_step
    try:
        out = ref_dd.getDistilData(net, 'imagenet', shape[0], num_batch=1, gpu=False, early_break_factor=0.0)
    finally:
        Plateau.__init__, Plateau.step = orig_init, orig_step
        del ref_dd.range
    assert len(out) == 1 and len(losses) == k
    rec = {'start': start.numpy().copy(), 'refined': out[0].numpy().copy(), 'losses': np.array(losses, dtype=np.float64),
           'cfg': np.array([seed, k, int(with_pixel_bn)])}
    for name, v in net.state_dict().items():
        rec['param.' + name] = v.numpy().copy()
    tag = 'zeroq_s{}{}'.format(seed, '_px' if with_pixel_bn else '')
    np.savez_compressed(os.path.join(GOLD, tag + '.npz'), **rec)
    print('{}: {} iterations, loss {:.4f} -> {:.4f}'.format(tag, k, losses[0], losses[-1]))


def main():
    run_case(0, 8, (4, 3, 16, 16), False)
    run_case(1, 8, (4, 3, 16, 16), True)


if __name__ == '__main__':
    main()
