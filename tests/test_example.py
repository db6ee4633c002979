"""examples/calibrate.py: the reference's main_cls.py calibration section end to end (every public pass once)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'examples'))


def test_calibration_section_end_to_end(engine, tmp_path, capsys):
    import calibrate
    table = str(tmp_path / 't.table')
    model, graph, bottoms = calibrate.main(['--net', 'tiny_mobile', '--table', table, '--device', str(engine.device)])
    out = capsys.readouterr().out
    assert 'equalisation sweeps' in out and 'distinct weight levels' in out
    lines = open(table).read().splitlines()
    n_layers = sum(hasattr(m, 'quant') for m in graph.values())
    assert len(lines) == 2 * n_layers
    for m in graph.values():
        if hasattr(m, 'quant'):
            assert float(m.quant.running_max) > float(m.quant.running_min)
            assert len(torch.unique(m.weight)) <= 256
    model(torch.randn(2, 3, 32, 32, device=engine.device))          # the quantised model still runs
