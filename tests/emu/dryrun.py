"""TEST INFRASTRUCTURE: run bench.py / __graft_entry__.smoke() logic on the CPU emulation (tiny net) to
catch Python-level mistakes before spending GPU minutes.   python tests/emu/dryrun.py"""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import torch
import build_emu
from dfq_amd import _ffi

_ffi._lib = _ffi.bind(ctypes.CDLL(build_emu.build()))
_ffi.target_device = lambda: torch.device('cpu')
_ffi.current_stream = lambda: 0
_ffi.synchronize = lambda: None

import bench
bench._device = lambda lr: torch.device('cpu')
bench._sync = lambda: None


def _elapsed(fn):
    import time
    t0 = time.perf_counter()
    fn()
    return (time.perf_counter() - t0) * 1e3


bench._gpu_elapsed_ms = _elapsed
import contextlib
bench._new_stream = lambda dev: None
bench._bind_thread = lambda dev: None
bench._stream_ctx = lambda s: contextlib.nullcontext()
world = int(os.environ.get('WORLD_SIZE', '1'))
sys.argv = ['bench.py', '--net', 'tiny_mobile', '--steps', '2', '--warmup', '1', '--cpu-seconds', '0.2', '--streams', '1', '--batch', '2',
            '--gpus', str(world), '--others', 'tiny_res,tiny_mobile:3', '--act-shape', '4,3,8,8', '--sharded', 'tiny_mobile:4,tiny_res:3',
            '--sharded-steps', '2', '--distill', 'tiny_mobile:2:4,3,16,16', '--pcie', 'tiny_mobile', '--lazy-steps', '1']
bench._BACKEND = 'gloo'            # the process group of bench.py over gloo (tests/test_bench_multirank.py)
if world > 1:
    # the multi-rank run checks the rank protocol (barrier, MAX over ranks, one line from rank 0) and the sharded pass; the
    # single-network legs are the single-rank run's business
    sys.argv = ['bench.py', '--net', 'tiny_mobile', '--steps', '2', '--warmup', '1', '--cpu-seconds', '0', '--streams', '1', '--batch', '2',
                '--gpus', str(world), '--others', '', '--act-shape', '', '--sharded', 'tiny_mobile:4,tiny_res:3', '--sharded-steps', '2',
                '--distill', '', '--pcie', '', '--lazy-steps', '0']
    bench.main()
    sys.exit(0)
bench.main()

# smoke() with cuda patched out
import __graft_entry__ as ge
torch.cuda.is_available = lambda: True
torch.cuda.synchronize = lambda *a, **k: None
_real_device = torch.device
class _Dev:
    def __call__(self, *a, **k):
        return _real_device('cpu')
import unittest.mock as mock
with mock.patch('torch.device', _Dev()):
    ge.smoke()
