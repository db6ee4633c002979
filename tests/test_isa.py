"""Compiler-metadata checks of every gfx950 kernel of the library (no GPU needed: hipcc cross-compiles).

DESIGN.md 4.5 tells the story of a kernel that silently kept a descriptor in scratch memory (2.5x slower); this test is the
promised guard: every kernel is compiled to gfx950 assembly (device side only, the flags of csrc/Makefile) and the
metadata the compiler emits per kernel is checked --

  * `.private_segment_fixed_size == 0` (no scratch memory, no vector-register spills) for every PRODUCTION kernel; the one
    opt-in kernel that has some (`le_sweep_kernel`, DFQ_LE_PERSIST=1, never the default path) is listed with a ceiling;
  * scalar-register spills (v_writelane / v_readlane moves, not memory) stay under per-kernel ceilings, so a change that
    makes a hot kernel spill more is seen at once;
  * no kernel uses MFMA (the path has no dense contraction: north_star) -- and the wave size is 64.
"""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'dfq_amd', 'csrc')
HIPCC = '/opt/rocm/bin/hipcc'
FLAGS = ['-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '--offload-arch=gfx950', '--cuda-device-only', '-x', 'hip', '-S',
         '-Wno-unused-command-line-argument', '-o', '-']

# kernels allowed to use scratch: name fragment -> (max scratch bytes, max vgpr spills)
SCRATCH_ALLOWED = {'le_sweep_kernel': (128, 110),          # opt-in persistent-workgroup variant (slower, kept for A/B)
                   'le_resident_kernelILb1': (32, 6)}       # the TUNING instantiation (trace stamps; tools/trace_*.py only) of a kernel at its register limit
# ceilings for scalar-register spills of the kernels that have any (everything else: 0)
SGPR_SPILL_CEILING = {                                      # (ILb0 = the production instantiation, ILb1 = the tuning one with trace stamps)
    'le_resident_kernel': 600, 'le_level_kernelILb0': 100, 'le_level_kernelILb1': 150, 'le_sweep_kernel': 120,
    'bc_chain_kernel': 115, 'bc_step_kernel': 60,
}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip') or f == 'dfq_core.cpp')


def _compile(src):
    res = subprocess.run([HIPCC] + FLAGS + [os.path.join(CSRC, src)], capture_output=True, text=True, cwd=CSRC)
    assert res.returncode == 0, res.stderr[-2000:]
    return src, res.stdout


@pytest.fixture(scope='module')
def kernels():
    if not os.path.exists(HIPCC):
        pytest.skip('no hipcc')
    out = []
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for src, asm in ex.map(_compile, _sources()):
            for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size:\s+\d+', asm, flags=re.S):
                blk = m.group(0)

                def g(k):
                    return re.search(r'\.%s:\s+(\S+)' % k, blk).group(1)
                out.append(dict(src=src, name=g('name'), vgpr=int(g('vgpr_count')), sgpr_spill=int(g('sgpr_spill_count')),
                                vgpr_spill=int(g('vgpr_spill_count')), scratch=int(g('private_segment_fixed_size')),
                                lds=int(g('group_segment_fixed_size')), wave=int(g('wavefront_size')), agpr=int(g('agpr_count'))))
            out.append(dict(src=src, name='__asm__', mfma=len(re.findall(r'^\s*v_mfma', asm, flags=re.M))))
    return out


def test_no_scratch_and_no_vector_spills(kernels):
    ks = [k for k in kernels if k['name'] != '__asm__']
    assert len(ks) >= 60, 'expected the whole library, got {} kernels'.format(len(ks))
    for k in ks:
        allowed = next((v for frag, v in SCRATCH_ALLOWED.items() if frag in k['name']), (0, 0))
        assert k['scratch'] <= allowed[0], '{} ({}): {} B of scratch'.format(k['name'], k['src'], k['scratch'])
        assert k['vgpr_spill'] <= allowed[1], '{} ({}): {} spilled vector registers'.format(k['name'], k['src'], k['vgpr_spill'])
        assert k['wave'] == 64


def test_scalar_spill_ceilings(kernels):
    for k in kernels:
        if k['name'] == '__asm__':
            continue
        ceiling = next((v for frag, v in SGPR_SPILL_CEILING.items() if frag in k['name']), 0)
        assert k['sgpr_spill'] <= ceiling, '{} ({}): {} scalar-register spills > {}'.format(k['name'], k['src'], k['sgpr_spill'], ceiling)


def test_no_mfma_anywhere(kernels):
    """north_star: "no MFMA (no dense contraction here)" -- HBM-bound reductions and rescales are not reshaped into GEMMs."""
    for k in kernels:
        if k['name'] == '__asm__':
            assert k['mfma'] == 0, '{}: {} MFMA instructions'.format(k['src'], k['mfma'])
        else:
            assert k['agpr'] == 0, k['name']


def test_occupancy_the_plans_count_on(kernels):
    """le_resident_kernel is planned at three workgroups of four waves per CU (<= 168 vector registers), le_level_kernel and
    the batch body of bc_chain_kernel at five to six (<= 80 / 88), the single-network body of the chain at four (<= 128)."""
    by = {}
    for k in kernels:
        if k['name'] != '__asm__':
            by.setdefault(k['name'], k)
    res = [k for n, k in by.items() if 'le_resident_kernel' in n]
    lev = [k for n, k in by.items() if 'le_level_kernel' in n]
    chain = [k for n, k in by.items() if 'bc_chain_kernel' in n]
    assert res and lev and chain
    assert all(k['vgpr'] <= 168 for k in res), [k['vgpr'] for k in res]
    assert all(k['vgpr'] <= 80 for k in lev), [k['vgpr'] for k in lev]
    # (round 4: float64 pdf / cdf code inlined into the chain's source-merge loop doubled its registers and cost 28 % of a batch's
    # correction time before anybody looked)
    # (round 5: two bodies.  A batch's is the lean one -- residency ahead of the chain's front is what its time is made of; a single
    # network's settles the row sum's operands before the wait and may take four waves' worth of registers)
    batch = [k for n, k in by.items() if 'bc_chain_kernel' in n and 'ELb0E' in n]
    single = [k for n, k in by.items() if 'bc_chain_kernel' in n and 'ELb1E' in n]
    assert batch and single and len(batch) + len(single) == len(chain)
    assert all(k['vgpr'] <= 88 for k in batch), [k['vgpr'] for k in batch]
    assert min(k['vgpr'] for k in batch) <= 80
    assert all(k['vgpr'] <= 128 for k in single), [k['vgpr'] for k in single]
