#!/bin/bash
# Runs on the GPU box (via gpurun): GPU test suite, the default bench line, and the rocprofv3 evidence
# for it.  Everything lands in gpurun_out/r01/ ; the summaries to keep are copied to profiles/ afterwards.
set -o pipefail
OUT=gpurun_out/r01
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== gpu tests"; timeout -k 10 600 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
echo "== bench (default flags)"; timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
echo "== bench --skip_wasted_D_wgrad"; timeout -k 10 300 python bench.py --skip_wasted_D_wgrad --no_cpu_baseline > $OUT/bench_skip.json 2>> $OUT/bench.err; echo "exit $?"
echo "== bench n_update_G=1"; timeout -k 10 300 python bench.py --n_update_G 1 --no_cpu_baseline > $OUT/bench_nug1.json 2>> $OUT/bench.err; echo "exit $?"
echo "== rocprofv3 kernel trace + stats of the bench command"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 50 --warmup 10 --no_cpu_baseline > $OUT/rocprof_stats.log 2>&1; echo "exit $?"
echo "== rocprofv3 PMC passes (eager, few steps)"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python tools/prof_step.py --steps 2 > $OUT/pmc_fetch.log 2>&1; echo "exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python tools/prof_step.py --steps 2 > $OUT/pmc_write.log 2>&1; echo "exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_mfma -- python tools/prof_step.py --steps 2 > $OUT/pmc_mfma.log 2>&1; echo "exit $?"
ls -R $OUT | head -50
cat $OUT/bench.json | cut -c1-600
