#!/bin/bash
# resident engine A/B: in-tree library vs variants/libdfq_hip_*.so (latency probe + phase trace each)
mkdir -p gpurun_out/abres
timeout 600 python -m pytest tests/test_engine_parity.py -m gpu -x -q -k "resident or full_size or engines_agree" > gpurun_out/abres/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/abres/pytest.log
for lib in dfq_amd/libdfq_hip.so variants/libdfq_hip_*.so; do
  [ -f $lib ] || continue
  tag=$(basename $lib .so | sed 's/libdfq_hip_\?//'); [ -z "$tag" ] && tag=base
  echo "== $tag"
  lazy=0; [[ $tag == *lazy* ]] && lazy=1
  DFQ_HIP_LIB=$PWD/$lib timeout 300 python tools/lat.py mobilenet_v2 deeplab_mnv2:60 2>/dev/null | tee gpurun_out/abres/lat_$tag.json
  DFQ_TRACE_LAZY=$lazy DFQ_HIP_LIB=$PWD/$lib timeout 300 python tools/trace_resident.py mobilenet_v2 8 > gpurun_out/abres/trace_$tag.txt 2>&1; tail -1 gpurun_out/abres/trace_$tag.txt
done
