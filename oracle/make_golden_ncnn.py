"""Pin the int8 calibration-table writer to the REFERENCE and write tests/golden/ncnn_table.json.

Runs only in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_ncnn.py

The reference's table writer is INLINE code of a script that cannot be imported (convert_ncnn.py needs torchvision, onnx
and a built ncnn tree): the block convert_ncnn.py:179-197 that turns `graph` + the table ncnn2table wrote (`table_old`)
into `table_new`.  Two things of the reference pin it:

  1. the table the reference HOLDS, modeling/ncnn/model_quant_relu_equal.table (106 lines: 53 `<name>_param_0 s x O`
     weight lines, then 53 `<name> s` activation lines) -- its first column, the number of tokens of every line, that a
     weight line repeats ONE scale and that every token is `str(float)` formatted;
  2. the block itself, EXECUTED here: its source lines are read from /root/reference/convert_ncnn.py at generation time
     (nothing of it is committed), dedented and run with `graph` = the synthetic MobileNetV2 of the bench (53 layers, the
     reference's own layer widths), `table_old` = the held table's lines, and seeded float32 activation ranges on every
     layer's `.quant` -- the lines it produces are the fixture (stored as name / scale string / repeat count per line).

The numpy oracle (`oracle.dfq_oracle.ncnn_table_lines`) is asserted string-identical to the executed block here; the
tests then hold `dfq_amd.ncnn_table` (one multi-tensor min/max launch on the device) to the same strings.
"""
from __future__ import annotations

import json
import os
import sys
import textwrap
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import torch                                   # noqa: E402
import torch.nn as nn                          # noqa: E402

from oracle import dfq_oracle as orc           # noqa: E402
from oracle import graphspec                   # noqa: E402
from dfq_amd import synthetic                  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
TABLE = os.path.join(REF, 'modeling', 'ncnn', 'model_quant_relu_equal.table')
SCRIPT = os.path.join(REF, 'convert_ncnn.py')
FIRST, LAST = 179, 197                         # convert_ncnn.py: `table_old = ...` up to `count += 1`
TARG = [nn.Conv2d, nn.Linear]


def activation_ranges(n_layers, seed=0):
    """Seeded float32 (running_min, running_max) per layer: what set_quant_minmax / update_quant_range leave in `.quant`."""
    rng = np.random.default_rng(seed)
    lo = -rng.uniform(0.0, 4.0, n_layers).astype(np.float32)
    hi = rng.uniform(0.5, 9.0, n_layers).astype(np.float32)
    lo[::5] = 0.0                              # layers behind a ReLU: min 0
    return lo, hi


def attach_ranges(graph, lo, hi, targ=TARG):
    keys = [k for k in graph if type(graph[k]) in targ]
    for i, k in enumerate(keys):
        q = types.SimpleNamespace(running_min=torch.tensor([float(lo[i])]), running_max=torch.tensor([float(hi[i])]))
        object.__setattr__(graph[k], 'quant', q)
    return keys


def reference_block():
    """Source of convert_ncnn.py:179-197, read where it lies; the first line is replaced by our own `table_old`."""
    lines = open(SCRIPT).read().split('\n')[FIRST - 1:LAST]
    assert lines[0].strip().startswith('table_old = ') and lines[-1].strip() == 'count += 1', (lines[0], lines[-1])
    return textwrap.dedent('\n'.join(lines[1:]))


def main():
    held = [line.rstrip('\n') for line in open(TABLE)]
    held = [line.strip() for line in held if line.strip()]
    names = [line.split(' ')[0] for line in held]
    counts = [len(line.split(' ')) - 1 for line in held]
    assert len(held) == 106 and all(n.endswith('_param_0') for n in names[:53]) and names[53:] == [n[:-8] for n in names[:53]]
    for line, c in zip(held[:53], counts[:53]):
        toks = line.split(' ')[1:]
        assert len(set(toks)) == 1 and c == len(toks)                 # ONE scale per layer, written once per output channel
    assert counts[53:] == [1] * 53
    for line in held:
        for t in set(line.split(' ')[1:]):
            assert str(float(t)) == t, t                              # str(float) formatting

    model, graph, bottoms = synthetic.build('mobilenet_v2', seed=0)
    lo, hi = activation_ranges(53)
    keys = attach_ranges(graph, lo, hi)
    assert [graph[k].weight.shape[0] for k in keys] == counts[:53], 'the synthetic MobileNetV2 has the held table\'s layer widths'

    ns = {'torch': torch, 'graph': graph, 'table_old': held}
    exec(compile(reference_block(), 'reference/convert_ncnn.py:180-197', 'exec'), ns)
    table_new = ns['table_new']
    assert len(table_new) == 106 and ns['count'] == 106

    spec = graphspec.from_torch(graph, bottoms, TARG)
    act = {k: (float(lo[i]), float(hi[i])) for i, k in enumerate(spec.targ_keys())}
    assert orc.ncnn_table_lines(spec, act, names=names) == table_new, 'oracle vs the executed reference block'

    out = {'what': 'oracle/make_golden_ncnn.py: convert_ncnn.py:180-197 executed on synthetic.build("mobilenet_v2", seed=0) with '
                   'the names of modeling/ncnn/model_quant_relu_equal.table and seeded activation ranges',
           'names': names, 'held_token_counts': counts,
           'act_min': [float(v) for v in lo], 'act_max': [float(v) for v in hi],
           'lines': [[l.split(' ')[0], l.split(' ')[1], len(l.split(' ')) - 1] for l in table_new]}
    for l, (n, s, c) in zip(table_new, out['lines']):
        assert l == ' '.join([n] + [s] * c)
    with open(os.path.join(GOLD, 'ncnn_table.json'), 'w') as f:
        json.dump(out, f, indent=0)
    print('ncnn_table.json: {} lines, weight-line token counts {} ... {}'.format(len(table_new), counts[:7], counts[51:53]))


if __name__ == '__main__':
    main()
