#!/bin/bash
# the PCIe-inclusive pass (CPU-resident model through the drop-in entry points) on its own, repeated, with the time of its
# staging steps (tuning aid)
mkdir -p gpurun_out/pcie
python - <<'PY' 2>/dev/null
import bench, torch, time, gc
from dfq_amd import _ffi, dfq
dev = bench._device(0)
T = {}
def timed(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[label or name] = T.get(label or name, 0.0) + (time.perf_counter() - t0) * 1e3; return r
    setattr(obj, name, g)
timed(_ffi, '_pinned'); timed(_ffi, '_to_device'); timed(_ffi, '_to_host')
timed(_ffi.Stage, 'prefetch'); timed(_ffi.Stage, 'writeback')
sync = torch.cuda.Stream.synchronize
def s2(self):
    t0 = time.perf_counter(); sync(self); T['stream_sync'] = T.get('stream_sync', 0.0) + (time.perf_counter() - t0) * 1e3
torch.cuda.Stream.synchronize = s2
for i in range(10):
    T.clear()
    r = bench.pcie_inclusive_pass('mobilenet_v2', reps=1)
    print(i, {k: round(v, 1) for k, v in r.items() if k in ('merge_batchnorm_ms', 'le_plus_bc_ms')}, {k: round(v, 1) for k, v in T.items()})
PY
