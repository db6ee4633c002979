"""Test configuration.

Markers
  gpu : needs a real MI355X (run by `pytest -m gpu` on the GPU box).  Everything else runs on CPU.

Backends (fixture ``engine``)
  'emu' : the *unmodified* kernel sources compiled with g++ against the fiber-based HIP emulation in
          tests/emu (TEST INFRASTRUCTURE; patched into dfq_amd._ffi for the duration of one test).
          Exercises kernel logic, the C ABI and the Python host layer without a GPU.
  'gpu' : the product path -- dfq_amd/libdfq_hip.so on cuda:0.  These are the parity tests proper.
"""
import ctypes
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X GPU (libdfq_hip.so on cuda:0)')


class Engine:
    def __init__(self, kind, device):
        self.kind = kind
        self.device = device

    def to(self, t):
        return t.to(self.device)


@pytest.fixture(scope='session')
def emu_lib_path():
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
    import build_emu
    return build_emu.build()


@pytest.fixture(params=['emu', pytest.param('gpu', marks=pytest.mark.gpu)])
def engine(request, monkeypatch):
    import torch
    from dfq_amd import _ffi
    if request.param == 'emu':
        path = request.getfixturevalue('emu_lib_path')
        handle = _ffi.bind(ctypes.CDLL(path))
        monkeypatch.setattr(_ffi, '_lib', handle)
        monkeypatch.setattr(_ffi, 'target_device', lambda: torch.device('cpu'))
        monkeypatch.setattr(_ffi, 'current_stream', lambda: 0)
        monkeypatch.setattr(_ffi, 'synchronize', lambda: None)
        yield Engine('emu', torch.device('cpu'))
    else:
        assert torch.cuda.is_available(), 'gpu-marked test needs a ROCm GPU'
        _ffi.lib()      # fails loudly if the HIP extension is missing
        yield Engine('gpu', torch.device('cuda', 0))
        torch.cuda.synchronize()
