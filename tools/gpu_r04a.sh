#!/bin/bash
# round 4, first GPU call: GPU suite on the regenerated DeepLab fixtures, smoke(), the default bench line (reference timed on this box)
mkdir -p gpurun_out/r04a
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04a/pytest.log 2>&1; echo "pytest rc=$? ${SECONDS}s"; grep -E "passed|failed" gpurun_out/r04a/pytest.log | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r04a/bench_default.json 2> gpurun_out/r04a/bench_default.err; echo "bench rc=$? ${SECONDS}s"
python tools/bench_line.py gpurun_out/r04a/bench_default.json | cut -c1-400
nproc; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04a/bench_default.json').read().strip().splitlines()[-1])
print(json.dumps(d['cpu_baseline'])[:1500])
print('others', json.dumps(d['config'].get('others'))[:800])
print('sharded', json.dumps(d.get('sharded'))[:600])
PY
