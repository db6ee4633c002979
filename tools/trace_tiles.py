import sys, json, torch, torch.nn as nn
sys.path.insert(0, '/root/repo')
import bench
from dfq_amd import dfq
dev = torch.device('cuda', 0)
proto = bench.prepare('mobilenet_v2', 0, dev)
for launch in range(5):
    rep = bench.make_replica(proto)
    info = rep['le'].level_info(launch)
    for block in sorted(set([0, info['workgroups'] // 2, info['workgroups'] - 1])):
        rep = bench.make_replica(proto)
        st = rep['le'].trace(launch, block)
        d = [st[i] - st[0] for i in range(8)]
        print('launch', launch, 'block', block, 'of', info['workgroups'], 'cycles since entry', d)
