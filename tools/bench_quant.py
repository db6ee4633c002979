#!/usr/bin/env python3
"""Throughput of the activation-side kernels (QuantMeasure of BASELINE.json config 5) on one large tensor."""
import sys
import time
import torch
sys.path.insert(0, '.')
from dfq_amd.utils import quantize as q

x = torch.randn(64, 96, 112, 112, device='cuda')
n = x.numel()


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


m = q.QuantMeasure(update_stat=True).cuda().eval()
dt = timeit(lambda: m(x))
print('QuantMeasure update_stat + quantise: %.3f ms, %.0f GB/s of 12 B/element' % (dt * 1e3, n * 12 / dt / 1e9))
m.set_update_stat(False)
dt = timeit(lambda: m(x))
print('QuantMeasure quantise only:          %.3f ms, %.0f GB/s of 8 B/element' % (dt * 1e3, n * 8 / dt / 1e9))
dt = timeit(lambda: q.quantize(x, 8, -2.0, 2.0))
print('quantize(x, 8, min, max):            %.3f ms, %.0f GB/s of 8 B/element' % (dt * 1e3, n * 8 / dt / 1e9))
dt = timeit(lambda: q.tensor_minmax(x))
print('tensor_minmax:                       %.3f ms, %.0f GB/s of 4 B/element' % (dt * 1e3, n * 4 / dt / 1e9))
