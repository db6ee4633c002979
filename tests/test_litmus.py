"""Hardware litmus test of the protocol the one-launch sweep and the one-launch correction chain rely on
(tools/litmus/xcd_flag.hip, built by __graft_entry__.build()): words published with device-scope atomics and a
counter bumped after `s_waitcnt 0` are read correctly by workgroups on other XCDs that spin on the counter and load
with device-scope loads -- without fences; the same exchange with plain stores / loads is NOT coherent."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, 'tools', 'litmus', 'xcd_flag')


def _run(mode, launches):
    out = subprocess.run([BIN, str(mode), str(launches)], capture_output=True, text=True, timeout=120, check=True).stdout
    m = re.search(r'(\d+) stale words, (\d+) give-ups', out)
    assert m, out
    return int(m.group(1)), int(m.group(2))


@pytest.mark.gpu
def test_device_scope_protocol_is_coherent_across_xcds():
    if not os.path.exists(BIN):
        pytest.skip('tools/litmus/xcd_flag not built (python -c "import __graft_entry__ as g; g.build()")')
    stale, gave_up = _run(0, 2000)
    assert stale == 0 and gave_up == 0
    stale_plain, _ = _run(1, 200)
    assert stale_plain > 0, 'plain stores/loads were expected to be incoherent across XCDs within a launch'


@pytest.mark.gpu
def test_register_file_butterfly_steps_match_shfl_xor():
    """xor_lane_minmax<1..32> (DPP / v_permlane swaps, dfq_common.hpp) == min / max with __shfl_xor, on the hardware."""
    binary = os.path.join(ROOT, 'tools', 'litmus', 'lane_xor')
    if not os.path.exists(binary):
        pytest.skip('tools/litmus/lane_xor not built (python -c "import __graft_entry__ as g; g.build()")')
    out = subprocess.run([binary], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and '0 mismatches' in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_every_xcd_dispatches_its_share_of_a_grid_in_index_order():
    """le_level_kernel (20 000 workgroups, far more than the chip holds) and bc_chain_kernel let a workgroup wait for workgroups
    with LOWER indices only; that cannot deadlock as long as every XCD dispatches its share of a 1-D grid in index order (the
    argument is in the header of tools/litmus/dispatch_order.hip) -- which HIP does not promise.  The
    program measures it on the hardware: every workgroup takes a ticket when it starts and notes its XCD; replaying the
    tickets XCD by XCD, no workgroup ever started while a predecessor further back than what one XCD holds at once had not,
    while the XCDs drift thousands of workgroups apart; and a chain -- every workgroup waiting for the one before it, the
    deepest dependency such waits allow -- runs through without a stall at up to 100 000 workgroups (200 000 by hand).  (The engine's waits are
    bounded and report DFQ_ERR_STATE anyway: tests/test_errors.py.)"""
    binary = os.path.join(ROOT, 'tools', 'litmus', 'dispatch_order')
    if not os.path.exists(binary):
        pytest.skip('tools/litmus/dispatch_order not built (python -c "import __graft_entry__ as g; g.build()")')
    for grid, threads, spin in ((21280, 256, 200), (100000, 256, 200), (4096, 1024, 200), (21280, 256, 0), (50000, 64, 50)):
        # (the ticket is taken by the workgroup's first instruction, some time after its dispatch: another process on the GPU
        # -- a test running next to this one under pytest-xdist -- can delay that arbitrarily, so a failed start-order verdict
        # is retried; the chain is the property itself and must hold every time)
        for attempt in range(3):
            out = subprocess.run([binary, str(grid), str(threads), str(spin)], capture_output=True, text=True, timeout=120)
            assert ' 0 stalled' in out.stdout and 'i mod n' in out.stdout, out.stdout + out.stderr
            if out.returncode == 0 and (spin == 0 or 'in index order' in out.stdout):
                break
        else:
            raise AssertionError(out.stdout + out.stderr)
