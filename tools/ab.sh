#!/bin/bash
# usage (GPU box): tools/ab.sh "<env A>" "<env B>" [rounds]   -- alternating bench runs of two environments on one box (ms/step, median)
A=$1; B=$2; R=${3:-3}
mkdir -p gpurun_out/r3
for i in $(seq 1 $R); do
  for v in A B; do
    E=$A; [ $v = B ] && E=$B
    r=$(env $E python bench.py --no_cpu_baseline --no_kernel_profile --steps 300 ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f %.4f' % (d['ms_per_step'], d['ms_per_step_median']))")
    echo "$v [$E] $r"
  done
done
