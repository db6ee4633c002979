#!/usr/bin/env python3
"""Digest the two rocprofv3 --pmc passes of tools/pmc_level.sh into a PMC summary (profiles/r01_pmc_summary.json).

usage: tools/pmc_digest.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/> <bench.json of the same --batch> <dest.json>

FETCH_SIZE is reported in KB with 128-byte requests tallied as 64 bytes on gfx950 (MI355X_MICROARCH.md,
HBM section) -> doubled for the wide coalesced reads of these kernels; bc_minmax_kernel (one plain read
of every weight of the batch) is the in-run calibration of that factor.  Only the le_level_kernel launches
of the timed batch are averaged (identified by their grid size: the process also runs one-network plans
for its latency probe); launches that exit at the `done` flag (< 64 KB moved) are excluded.
"""
import collections
import csv
import glob
import json
import sys


def load(counter, root):
    files = glob.glob('%s/pmc_%s/**/*counter_collection.csv' % (root, counter), recursive=True)
    if not files:
        raise SystemExit('no counter file for ' + counter)
    return [r for r in csv.DictReader(open(files[0])) if r.get('Counter_Name') == counter]


def main():
    root, bench_path, dest = sys.argv[1], sys.argv[2], sys.argv[3]
    bench = json.loads(open(bench_path).read().splitlines()[0])
    levels = bench['roofline']['levels']
    grid_of = {l['workgroups'] * 256: l for l in levels}
    batch = bench['config']['networks_per_step']
    out = {'batch': batch, 'workload': bench['config']['workload']}
    per_level = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        rows = load(c, root)
        by_kernel = collections.defaultdict(list)
        for r in rows:
            by_kernel[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
        out[c] = {'per_kernel_mean_KB': {k: sum(v) / len(v) for k, v in by_kernel.items() if 'dfq' in k},
                  'per_kernel_dispatches': {k: len(v) for k, v in by_kernel.items() if 'dfq' in k}}
        for r in rows:
            if 'le_level_kernel' in r['Kernel_Name'] and int(r['Grid_Size']) in grid_of and float(r['Counter_Value']) > 64.0:
                per_level.setdefault(int(r['Grid_Size']), {}).setdefault(c, []).append(float(r['Counter_Value']))
    table = []
    tot_fetch = tot_write = tot_alg = 0.0
    n_levels = 0
    for grid, l in sorted(grid_of.items(), key=lambda kv: kv[1]['level']):
        if grid not in per_level or len(per_level[grid]) < 2:
            continue
        f = per_level[grid]['FETCH_SIZE']
        w = per_level[grid]['WRITE_SIZE']
        fetch = sum(f) / len(f) * 1024 * 2.0
        write = sum(w) / len(w) * 1024
        table.append({'level': l['level'], 'workgroups': l['workgroups'], 'launches_counted': len(f),
                      'fetch_bytes_corrected': fetch, 'write_bytes': write, 'traffic_bytes': fetch + write,
                      'algorithmic_bytes': l['bytes'], 'traffic_over_algorithmic': (fetch + write) / l['bytes']})
        tot_fetch += fetch
        tot_write += write
        tot_alg += l['bytes']
        n_levels += 1
    out['le_level_kernel'] = {
        'levels': table,
        'fetch_bytes_per_launch_corrected': tot_fetch / n_levels, 'write_bytes_per_launch': tot_write / n_levels,
        'traffic_bytes_per_launch': (tot_fetch + tot_write) / n_levels, 'algorithmic_bytes_per_launch': tot_alg / n_levels,
        'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc_level.sh --batch %d); FETCH_SIZE '
                'doubled per MI355X_MICROARCH.md; per-launch figures are the mean over the %d launch levels of a sweep, like '
                'bench.py roofline.bytes_per_launch' % (batch, n_levels),
    }
    # the second streaming kernel of a sweep (round 6): le_lean_kernel, one launch per group of sweeps over the free-running layers
    fr = bench['roofline'].get('free_running')
    if fr:
        lean = {}
        for c in ('FETCH_SIZE', 'WRITE_SIZE'):
            v = [float(r['Counter_Value']) for r in load(c, root) if 'le_lean_kernel' in r['Kernel_Name'] and float(r['Counter_Value']) > 64.0]
            if v:
                lean[c] = (sum(v) / len(v), len(v))
        if len(lean) == 2:
            fetch, write = lean['FETCH_SIZE'][0] * 1024 * 2.0, lean['WRITE_SIZE'][0] * 1024
            out['le_lean_kernel'] = {'launches_counted': lean['FETCH_SIZE'][1], 'fetch_bytes_corrected': fetch, 'write_bytes': write,
                                     'traffic_bytes_per_launch': fetch + write, 'algorithmic_bytes_per_launch': fr['bytes_per_launch'],
                                     'traffic_over_algorithmic': (fetch + write) / fr['bytes_per_launch'],
                                     'sweeps_per_launch': fr['sweeps_per_launch']}
    json.dump(out, open(dest, 'w'), indent=1)
    print(json.dumps(out['le_level_kernel'], indent=1))
    if 'le_lean_kernel' in out:
        print(json.dumps(out['le_lean_kernel'], indent=1))


if __name__ == '__main__':
    main()
