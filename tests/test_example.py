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


def test_calibration_section_with_distilled_ranges(engine, capsys):
    """--distill-range (BASELINE config 5, main_cls.py:86-113 + :183-186): ZeroQ batches -> update_quant_range."""
    import calibrate
    model, graph, bottoms = calibrate.main(['--net', 'tiny_mobile', '--distill-range', '--dis-iterations', '3', '--dis-num-batch', '2',
                                            '--dis-batch-size', '2', '--device', str(engine.device)])
    first = [k for k in graph if bottoms[k] is not None and bottoms[k][0] == 'Data' and hasattr(graph[k], 'quant')]
    assert len(first) == 1
    for k, m in graph.items():
        if hasattr(m, 'quant'):
            assert not m.quant.update_stat
            if k in first:       # pinned to the normalised image range (improve_dfq.py:290-291)
                assert abs(float(m.quant.running_max) - 2.64) < 1e-6 and abs(float(m.quant.running_min) + 2.11790393) < 1e-6
            else:
                assert float(m.quant.running_max) >= float(m.quant.running_min)
    assert any(float(m.quant.running_max) > 0 for m in graph.values() if hasattr(m, 'quant'))
    model(torch.randn(2, 3, 32, 32, device=engine.device))
