#!/bin/bash
# usage: tools/r3_run.sh <tag> [pytest -k expr | "all" | "none"]   -> gpurun_out/r3/<tag>.*  (tests, then the bench with its per-call table)
T=$1; K=${2:-all}
mkdir -p gpurun_out/r3
if [ "$K" = all ]; then timeout -k 10 600 python -m pytest -m gpu -x -q > gpurun_out/r3/$T.test.log 2>&1; rc=$?
elif [ "$K" != none ]; then timeout -k 10 600 python -m pytest -m gpu -x -q -k "$K" > gpurun_out/r3/$T.test.log 2>&1; rc=$?; else rc=0; fi
[ "$K" != none ] && tail -4 gpurun_out/r3/$T.test.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR|E  )" gpurun_out/r3/$T.test.log | head -20; exit $rc; }
SGAN_BENCH_CALLS=gpurun_out/r3/$T.calls.txt timeout -k 10 300 python bench.py --no_cpu_baseline ${BENCH_ARGS} > gpurun_out/r3/$T.json 2> gpurun_out/r3/$T.err || { tail -5 gpurun_out/r3/$T.err; exit 1; }
python - <<PY
import json
d=json.load(open('gpurun_out/r3/$T.json'))
print('ms/step', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median'],4), 'img/s', round(d['value'],1), 'dom', d['roofline']['kernel'], round(d['roofline']['achieved'],1),'TF')
PY
