"""bench.py's N > 1 path (one rank per GPU, barrier + MAX over ranks, rank 0 prints the line) on two CPU
ranks over gloo, with the kernels running on the CPU emulation (tests/emu/dryrun.py).  The driver launches
the real thing as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` over RCCL."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_two_ranks_weak_scaling_line():
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), PYTHONPATH=ROOT)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'emu', 'dryrun.py')], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith('{"metric"')]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['config']['networks_per_step'] == 4
    assert d['value'] > 0 and d['vs_baseline'] is None and 'cpu_baseline' not in d     # CPU baseline: N=1 only
    # config 4 as north_star splits it: both ranks on the data path, one collective per pass, strong scaling
    assert [e['net'] for e in d['sharded']] == ['tiny_mobile', 'tiny_res']
    for sh in d['sharded']:
        assert sh['world'] == 2 and sh['ranks_owning_components'] == 2 and sh['collectives_per_pass'] == 1
        assert sh['scaling'] == 'strong' and sh['sweeps_pinned'] and sh['value'] > 0 and sh['exchange_bytes_per_rank'] > 0
        # who holds what: greedy split of the components by paired elements
        assert len(sh['paired_elements_per_rank']) == 2 and all(n > 0 for n in sh['paired_elements_per_rank'])
        assert sh['components'] >= 2
        # the reference's own stopping rule across ranks: one all_reduce per sweep + the final all_gather
        assert sh['data_dependent_ms'] > 0 and sh['data_dependent_sweeps'] >= 1
        assert sh['data_dependent_collectives_per_pass'] == -(-sh['data_dependent_sweeps'] // sh['data_dependent_chunk']) + 1   # one all_reduce per chunk of sweeps (sharded.py) + the all_gather


def test_bench_line_contract_single_rank():
    """The JSON line of bench.py (dry run on the emulator, tiny net): every field the measurement contract names."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'emu', 'dryrun.py')], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['unit'] == 'weights/s' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and d['n_gpus'] == 1
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in r, key
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    c = d['cpu_baseline']
    for key in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert key in c, key
    assert c['kind'] == 'port' and c['cores'] == 1 and c['unit'] == 'weights/s'
    assert abs(d['value'] - d['config']['networks_per_step'] * 5000 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    assert 'frac_survey_8d' not in r and 'eager_formulation_equivalent_GBps' in r     # not a fraction
    lat = d['latency']
    assert lat['single_network_pass_ms'] > 0 and 0 < lat['frac_of_hbm_peak'] and lat['sweeps'] == d['config']['le_sweeps'][0]
    others = d['config']['others']
    assert [o['net'] for o in others] == ['tiny_res', 'tiny_mobile'] and others[1]['sweeps_pinned'] and others[1]['sweeps'] == 3
    assert all(o['ms'] > 0 and o['roofline_frac'] > 0 for o in others)
    act = d['config']['activation_ranges']
    assert [k['bytes'] for k in act['kernels'][:3]] == [4 * act['elements'], 8 * act['elements'], 12 * act['elements']]
    assert [e['net'] for e in d['sharded']] == ['tiny_mobile', 'tiny_res']
    assert all(e['world'] == 1 and e['scaling'] == 'strong' and e['data_dependent_ms'] > 0 for e in d['sharded'])
    # one rank, same network: the data-dependent sharded loop stops where the plain entry point's loop stops
    assert d['sharded'][0]['data_dependent_sweeps'] == d['config']['le_sweeps'][0]
    # like-for-like figure, config 5 end to end, the PCIe-inclusive drop-in pass
    assert abs(d['value_single_network'] - lat['weights_per_s']) < 1e-6 * lat['weights_per_s']
    dr = d['config']['distill_range']
    assert dr['batches'] == 2 and dr['quant_measures'] > 0 and dr['ms_per_batch'] > 0
    assert dr['quant_measure_bytes_per_batch'] == 12 * dr['elements_per_batch'] and dr['elements_per_batch'] > 0
    assert d['pcie_inclusive_ms'] == d['pcie_inclusive']['le_plus_bc_ms'] > 0
    lz = d['lazy_scale']            # the opt-in formulation: its own value and its own roofline entry, never `value`
    assert lz['sweeps'] == d['config']['le_sweeps'] and lz['sweeps_given'] and d['value_lazy_scale'] == lz['value'] > 0
    assert abs(lz['roofline']['frac'] - lz['roofline']['achieved'] / lz['roofline']['peak']) < 1e-12
    assert 'reference' in c          # the committed reference-CPU record (null for nets it was not measured on)
