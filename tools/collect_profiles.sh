#!/bin/bash
# Runs on the GPU box (via gpurun): the bench lines and the rocprofv3 evidence for them.  Everything lands in
# gpurun_out/<round>/ ; tools/summarize_profiles.py copies the summaries to keep into profiles/.
set -o pipefail
R=${ROUND:-r03}
OUT=gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== bench (default flags)"; timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
echo "== bench --skip_wasted_D_wgrad"; timeout -k 10 300 python bench.py --skip_wasted_D_wgrad --no_cpu_baseline > $OUT/bench_skip.json 2>> $OUT/bench.err; echo "exit $?"
echo "== bench n_update_G=1"; timeout -k 10 300 python bench.py --n_update_G 1 --no_cpu_baseline > $OUT/bench_nug1.json 2>> $OUT/bench.err; echo "exit $?"
echo "== bench --eager"; timeout -k 10 300 python bench.py --eager --no_cpu_baseline --no_kernel_profile > $OUT/bench_eager.json 2>> $OUT/bench.err; echo "exit $?"
echo "== bench --workload cgan"; timeout -k 10 400 python bench.py --workload cgan --steps 50 --warmup 5 > $OUT/bench_cgan.json 2>> $OUT/bench.err; echo "exit $?"
echo "== bench --workload twostage_cycle"; timeout -k 10 300 python bench.py --workload twostage_cycle --steps 30 --warmup 5 > $OUT/bench_twostage.json 2>> $OUT/bench.err; echo "exit $?"
echo "== rocprofv3 kernel trace + stats of the bench command (hipGraph replay)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 50 --warmup 10 --no_cpu_baseline > $OUT/rocprof_stats.log 2>&1; echo "exit $?"
echo "== rocprofv3 kernel trace + stats, eager single stream (the regime of bench.py's live per-kernel measurement)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_eager -- python bench.py --eager --no_d_streams --steps 20 --warmup 5 --no_cpu_baseline > $OUT/rocprof_eager.log 2>&1; echo "exit $?"
echo "== same for the secondary workloads (their roofline objects quote the 64x64 kernel)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cgan_eager -- python bench.py --workload cgan --eager --no_d_streams --steps 10 --warmup 3 --no_cpu_baseline > $OUT/rocprof_cgan_eager.log 2>&1; echo "exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_twostage_eager -- python bench.py --workload twostage_cycle --eager --no_d_streams --steps 10 --warmup 3 --no_cpu_baseline > $OUT/rocprof_twostage_eager.log 2>&1; echo "exit $?"
echo "== rocprofv3 PMC passes (separate runs, eager, few steps)"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_FETCH_SIZE -- python tools/prof_step.py --steps 2 --no_d_streams > $OUT/pmc_fetch.log 2>&1; echo "exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_WRITE_SIZE -- python tools/prof_step.py --steps 2 --no_d_streams > $OUT/pmc_write.log 2>&1; echo "exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma -- python tools/prof_step.py --steps 2 --no_d_streams > $OUT/pmc_mfma.log 2>&1; echo "exit $?"
ls $OUT | head -40
cut -c1-400 $OUT/bench.json
