"""The C-ABI boundary: include/dfq_hip.h, the ctypes signature table and the built shared object must
agree symbol for symbol; the library must load without a GPU (no compute calls here)."""
import ctypes
import os
import re
import subprocess

import pytest

from dfq_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'dfq_hip.h')


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dfq_[a-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def product_lib():
    if not os.path.exists(_ffi.LIB_PATH):
        subprocess.run(['make', '-C', os.path.join(ROOT, 'dfq_amd', 'csrc'), '-j', '8'], check=True)
    return ctypes.CDLL(_ffi.LIB_PATH)


def test_header_matches_binding_table():
    assert header_functions() == sorted(_ffi.SIGNATURES)


def test_library_exports_every_symbol(product_lib):
    for name in header_functions():
        assert hasattr(product_lib, name), 'libdfq_hip.so does not export ' + name
    out = subprocess.run(['nm', '-D', '--defined-only', _ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r' T (dfq_[a-z0-9_]+)$', out, flags=re.M)))
    assert exported == header_functions(), 'exported C symbols differ from the header'


def test_library_is_gfx950_code_object(product_lib):
    blob = open(_ffi.LIB_PATH, 'rb').read()
    assert b'gfx950' in blob, 'no gfx950 code object embedded'
    assert b'le_level_kernel' in blob and b'bc_step_kernel' in blob and b'fake_quant_kernel' in blob


def test_version_and_error_plumbing(product_lib):
    lib = _ffi.bind(product_lib)
    assert lib.dfq_version() == 100
    # argument validation happens before any HIP call, so it is testable without a GPU
    rc = lib.dfq_fake_quant(None, None, 10, 8, 0, 0, 0.0, 1.0, None, None, None)
    assert rc == -1
    assert b'dfq_fake_quant' in lib.dfq_last_error()
    rc = lib.dfq_le_plan_create(None, 0, None, 0, None)
    assert rc == -1


def test_struct_layouts_match_header():
    # sizes the C compiler gives the structs (natural alignment), checked against ctypes
    src = r'''
    #include <stdio.h>
    #include "dfq_hip.h"
    int main(void) {
        printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(dfq_layer), sizeof(dfq_relation), sizeof(dfq_le_config),
               sizeof(dfq_le_result), sizeof(dfq_segment), sizeof(dfq_bc_source), sizeof(dfq_bc_step), sizeof(dfq_rebuild_item));
        return 0;
    }'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, 't.c')
        open(c, 'w').write(src)
        exe = os.path.join(d, 't')
        subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe], check=True)
        sizes = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    want = [ctypes.sizeof(t) for t in (_ffi.DfqLayer, _ffi.DfqRelation, _ffi.DfqLeConfig, _ffi.DfqLeResult,
                                       _ffi.DfqSegment, _ffi.DfqBcSource, _ffi.DfqBcStep, _ffi.DfqRebuildItem)]
    assert sizes == want


def test_missing_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError, match='no ROCm GPU'):
        _ffi.Stage()
