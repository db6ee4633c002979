import sys, json, torch, torch.nn as nn
sys.path.insert(0, '/root/repo')
import bench
from dfq_amd import dfq
dev = torch.device('cuda', 0)
proto = bench.prepare('mobilenet_v2', 0, dev)
import os
print('ablate', os.environ.get('DFQ_LE_ABLATE'))
for launch in (1, 4):
    rep = bench.make_replica(proto)
    info = rep['le'].level_info(launch)
    gx, gy = info['grid']
    for block in sorted(set([0, (gy // 2) * gx, (gy - 1) * gx])):
        rep = bench.make_replica(proto)
        st = rep['le'].trace(launch, block)
        d = [st[i] - st[0] for i in range(8)]
        print('launch', launch, 'block', block, 'grid', info['grid'], 'working', info['workgroups'], 'cycles since entry', d)
