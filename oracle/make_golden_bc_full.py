"""Bias correction STAGE-WISE at BASELINE size against the UNMODIFIED reference: tests/golden/bcfull_*.npz.

Runs only in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_bc_full.py

The reference's bias_correction (dfq.py:173-293) takes half a second at full size; what takes 75 s is its equalisation
loop.  A stage-wise check does not need THAT loop's output, only a common, reproducible input state: the synthetic network
(seed 0) -> BN folded and equalised by the numpy oracle (deterministic: the GPU box reproduces it bit for bit; the fixture
stores float64 moments of every input tensor so that a differing input is reported as such).  From that state:

    reference  dfq.bias_correction(graph, bottoms, targ_type)      -> every layer bias, every BN proxy beta~ (stored)
    oracle     dfq_oracle.bias_correction(spec)                    -> asserted within 1e-5 here

tests/test_full_reference.py then runs the ENGINE from the same state (CPU: emulated kernels, GPU: the product library) and
compares with the stored reference outputs at 1e-5: the full-size, stage-wise BC parity in one committed test.
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)
sys.dont_write_bytecode = True

import torch                                   # noqa: E402
import torch.nn as nn                          # noqa: E402

import dfq as ref_dfq                          # noqa: E402  (reference)

from oracle import bc_full_state               # noqa: E402
from oracle import dfq_oracle as orc           # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
TARG = [nn.Conv2d, nn.Linear]


def run(name, sweeps):
    model, graph, bottoms, spec = bc_full_state.build(name, sweeps)
    out = {'sweeps': np.array(sweeps if sweeps else -1)}
    out.update(bc_full_state.input_moments(spec))
    with torch.no_grad():
        ref_dfq.bias_correction(graph, bottoms, TARG)            # the reference, on torch modules holding that state
    orc.bias_correction(spec)
    worst = 0.0
    keys = list(graph.keys())
    for i, k in enumerate(keys):
        n = spec.nodes[k]
        m = graph[k]
        if n.kind == 'targ' and m.bias is not None:
            ref = m.bias.detach().numpy().astype(np.float32)
            out['bc.L{}.b'.format(i)] = ref
            if n.bias is not None:
                err = np.abs(n.bias.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))
                worst = max(worst, float(err.max()))
        elif n.kind == 'bn' and n.fake_weight is not None:
            ref = m.fake_bias.detach().numpy().astype(np.float32)
            out['bc.L{}.fb'.format(i)] = ref
            err = np.abs(n.fake_bias.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))
            worst = max(worst, float(err.max()))
    assert worst <= 1e-5, '{}: oracle vs reference BC differ by {}'.format(name, worst)
    np.savez_compressed(os.path.join(GOLD, 'bcfull_{}.npz'.format(name)), **out)
    print('bcfull_{}: {} corrected vectors, oracle vs reference max rel err {:.2e}'.format(
        name, len([k for k in out if k.startswith('bc.')]), worst))


if __name__ == '__main__':
    os.makedirs(GOLD, exist_ok=True)
    run('mobilenet_v2', None)
    run('resnet18', None)
    run('deeplab_mnv2', 12)
