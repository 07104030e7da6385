#!/bin/bash
# round-3 first GPU call: suite on the new operand checks, graph-overlap probe under the runtime's graph knobs, baseline bench
mkdir -p gpurun_out/r3
python -m pytest -m gpu -x -q > gpurun_out/r3/t1.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3/t1.log
tail -3 gpurun_out/r3/t1.log
for v in "" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_HIP_FORCE_GRAPH_QUEUES=4" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=4" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"; do
  echo "== env: $v" >> gpurun_out/r3/overlap.log
  env $v timeout -k 10 120 python tools/graph_overlap_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3/overlap.log
done
cat gpurun_out/r3/overlap.log
timeout -k 10 300 python bench.py > gpurun_out/r3/bench0.json 2> gpurun_out/r3/bench0.err; echo "bench exit $?"
python -c "import json;d=json.load(open('gpurun_out/r3/bench0.json'));print(d['value'],d['ms_per_step'],d['ms_per_step_median'],d['roofline']['frac'])"
