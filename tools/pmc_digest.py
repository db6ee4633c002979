#!/usr/bin/env python3
"""Digest the two rocprofv3 --pmc passes of tools/pmc_level.sh into profiles/r01_pmc_summary.json.

FETCH_SIZE is reported in KB with 128-byte requests tallied as 64 bytes on gfx950 (MI355X_MICROARCH.md,
HBM section) -> doubled for the wide coalesced reads of these kernels; bc_minmax_kernel (one plain read
of every weight) is the in-run calibration of that factor.  Launches of le_level_kernel that exit at the
`done` flag (sweeps enqueued past convergence) move < 64 KB and are excluded from the per-launch mean.
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(counter, root):
    files = glob.glob('%s/pmc_%s/**/*counter_collection.csv' % (root, counter), recursive=True)
    if not files:
        raise SystemExit('no counter file for ' + counter)
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r.get('Counter_Name') == counter:
            rows[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    return rows


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out'
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    alg = float(sys.argv[3]) if len(sys.argv) > 3 else None
    out = {'batch': batch}
    level = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        rows = per_kernel(c, root)
        out[c] = {'per_kernel_mean_KB': {k: sum(v) / len(v) for k, v in rows.items()},
                  'per_kernel_dispatches': {k: len(v) for k, v in rows.items()}}
        lv = rows.get('dfq::le_level_kernel', [])
        work = [x for x in lv if x > 64.0]
        level[c] = (sum(work) / max(len(work), 1), len(work), len(lv))
    fetch = level['FETCH_SIZE'][0] * 1024 * 2.0
    write = level['WRITE_SIZE'][0] * 1024
    out['le_level_kernel'] = {
        'working_launches': level['FETCH_SIZE'][1], 'launches': level['FETCH_SIZE'][2],
        'fetch_bytes_per_launch_corrected': fetch, 'write_bytes_per_launch': write,
        'traffic_bytes_per_launch': fetch + write, 'algorithmic_bytes_per_launch': alg,
        'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc_level.sh); '
                'FETCH_SIZE doubled per MI355X_MICROARCH.md; launches that exit at the done flag excluded',
    }
    json.dump(out, open('profiles/r01_pmc_summary.json', 'w'), indent=1)
    print(json.dumps(out['le_level_kernel']))


if __name__ == '__main__':
    main()
