#!/bin/bash
# kernel sequence of ResNet-18's single-network pass (streaming engine): durations and start-to-start gaps of the last pass
bash tools/gpu_prof.sh r18 -- python tools/lat.py resnet18 > gpurun_out/r18_digest.txt 2>&1
t=$(find gpurun_out/r18 -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
# last occurrence of the reset kernel starts the last LE pass
idx = [i for i, r in enumerate(rows) if 'le_prepare' in r['Kernel_Name'] or 'le_reset' in r['Kernel_Name']]
start = idx[-1]
t0 = int(rows[start]['Start_Timestamp'])
prev_end = t0
for r in rows[start:start + 40]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%8.1f us  +gap %6.1f  dur %7.1f  grid %8s  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r['Kernel_Name'][:60]))
    prev_end = e
PY
