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
