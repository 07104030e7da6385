#!/bin/bash
set -o pipefail
R=${ROUND:-r02}
OUT=gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== eager single-stream bench under rocprofv3 (regime of the live per-kernel measurement)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_eager -- python bench.py --eager --no_d_streams --steps 20 --warmup 5 --no_cpu_baseline > $OUT/bench_eager.json 2> $OUT/rocprof_eager.log; echo "exit $?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python tools/prof_step.py --steps 2 --no_d_streams > $OUT/pmc_$c.log 2>&1; echo "$c exit $?"
done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python tools/prof_step.py --steps 2 --no_d_streams > $OUT/pmc_mfma.log 2>&1; echo "mfma exit $?"
ls -R $OUT | grep -E "csv" | head -30
tail -2 $OUT/pmc_FETCH_SIZE.log | cut -c1-200
