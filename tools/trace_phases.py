#!/usr/bin/env python3
"""Phase stamps of single workgroups of the equalisation launches under full load (tuning aid, GPU box).
usage: tools/trace_phases.py [batch]   -> per sampled workgroup: ns since entry (shader clock at ~2.1 GHz) at the phase boundaries
[descriptor+state loaded, data loads issued, scales solved, barrier passed, elements stored, stats published, partial written]"""
import sys
import torch
sys.path.insert(0, '.')
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0)
protos = [bench.prepare('mobilenet_v2', seed=i, dev=dev) for i in range(batch)]
unit = bench.make_unit(protos)
for launch in range(unit['le'].levels):
    unit = bench.make_unit(protos)
    info = unit['le'].level_info(launch)
    n = info['grid'][0] * info['grid'][1]
    for block in sorted(set([0, n // 9, n // 7, n // 5, n // 3, n // 2, (3 * n) // 5, (2 * n) // 3, (4 * n) // 5, n - 1])):
        unit = bench.make_unit(protos)
        st = unit['le'].trace(launch, block)
        d = [round((st[i] - st[0]) / 2.1) if st[i] else -1 for i in range(1, 8)]   # ~2.1 GHz shader clock -> ns
        bi = unit['le'].block_info(launch, block)
        print('launch %d block %5d of %5d kind %d %4dx%-4d rw %6d ro %6d: ns since entry %s' % (launch, block, n, bi['kind'], bi['rows'], bi['cols'], bi['rw_elements'], bi['ro_elements'], d))
