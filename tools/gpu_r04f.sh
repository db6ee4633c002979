#!/bin/bash
# round 4: one-launch QuantMeasure (config 5) -- parity, kernel throughput on the largest activation, the distilled batches A/B
tag=r04f
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_engine_parity.py tests/test_range_parity.py tests/test_errors.py -m gpu -x -q -k "quant_measure or range or update_quant or abandoned" > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/$tag/pytest.log
for rep in 1 2; do
for f in 1 0; do
  echo "fused=$f"; DFQ_QM_FUSED=$f timeout 300 python tools/bench_quant.py 2>/dev/null | head -1
  DFQ_QM_FUSED=$f timeout 300 python tools/distill_probe.py 2>/dev/null | tail -1 | tee -a gpurun_out/$tag/distill_fused$f.json
done; done
