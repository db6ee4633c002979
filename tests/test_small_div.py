"""small_div (dfq_amd/csrc/dfq_le.hip) must be exact on its whole domain: checked here with numpy's
float32 arithmetic and a reciprocal perturbed by the 1-ulp error v_rcp_f32 is allowed to have."""
import numpy as np


def small_div(a, b, ulp_off):
    r = (np.float32(1.0) / b.astype(np.float32)).astype(np.float32)
    r = np.nextafter(r, np.float32(np.inf) * np.float32(ulp_off)) if ulp_off else r
    return ((a.astype(np.float32) + np.float32(0.5)) * r).astype(np.float32).astype(np.int64)


def test_small_div_exact():
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.integers(0, 1 << 20, 2_000_000), np.arange(0, 4096), np.arange((1 << 20) - 4096, 1 << 20)])
    b = np.concatenate([rng.integers(1, 1 << 20, 1_000_000), rng.integers(1, 300, 1_000_000),
                        rng.integers(1, 300, 8192)])[:a.size]
    # multiples of b and their neighbours are the critical cases
    k = rng.integers(0, 4096, a.size)
    a2 = np.minimum(k * b + rng.integers(-1, 2, a.size), (1 << 20) - 1).clip(0)
    for aa in (a, a2):
        for off in (0, 1, -1):
            assert np.array_equal(small_div(aa, b, off), aa // b)
