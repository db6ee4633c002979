"""tuning aid: how long the HOST takes to enqueue one step (LE graph + BC graph) vs the GPU time"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device('cuda', 0)
proto = bench.prepare('mobilenet_v2', 0, dev)
reps = [bench.make_replica(proto) for _ in range(12)]
sweeps = 47
def step(r):
    r['le'].enqueue(sweeps, restart=True, max_sweeps=sweeps)
    r['bc'].run()
for r in reps[:4]:
    step(r)
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in reps[4:]:
    step(r)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
n = len(reps) - 4
print('host enqueue per step %.3f ms, total per step %.3f ms' % ((t1 - t0) * 1e3 / n, (t2 - t0) * 1e3 / n))
# LE only / BC only GPU time
for name, fn in (('le', lambda r: r['le'].enqueue(sweeps, restart=True, max_sweeps=sweeps)), ('bc', lambda r: r['bc'].run()),
                 ('restart', lambda r: r['le'].enqueue(0, restart=True, max_sweeps=sweeps))):
    fresh = [bench.make_replica(proto) for _ in range(6)]
    for r in fresh[:2]:
        fn(r)
    torch.cuda.synchronize()
    ms = bench._gpu_elapsed_ms(lambda: [fn(r) for r in fresh[2:]]) / 4
    print(name, 'gpu ms per call %.3f' % ms)
# same on a created (non-null) stream
s = torch.cuda.Stream(device=dev)
reps = [bench.make_replica(proto) for _ in range(12)]
torch.cuda.synchronize()
with torch.cuda.stream(s):
    for r in reps[:4]:
        step(r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in reps[4:]:
        step(r)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print('non-null stream: host enqueue per step %.3f ms, total per step %.3f ms' % ((t1 - t0) * 1e3 / n, (t2 - t0) * 1e3 / n))
# threads: one python thread per stream
import threading
def worker(stream, mine):
    with torch.cuda.stream(stream):
        for r in mine:
            step(r)
for nthr in (2, 4, 8):
    reps = [bench.make_replica(proto) for _ in range(8 * nthr)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nthr)]
    torch.cuda.synchronize()
    ths = [threading.Thread(target=worker, args=(streams[i], reps[i::nthr][:2])) for i in range(nthr)]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(streams[i], reps[i::nthr][2:])) for i in range(nthr)]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('threads', nthr, 'ms per step %.3f' % ((t2 - t0) * 1e3 / (6 * nthr)))
