#!/bin/bash
# Re-runs only the bench lines (profiles/<round>_pmc_traffic.json from collect_profiles.sh + summarize_profiles.py must exist,
# bench.py reads the dominant kernel's HBM traffic from it).
set -o pipefail
R=${ROUND:-r03}
OUT=gpurun_out/$R
mkdir -p $OUT
echo "== bench (default flags)"; timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
echo "== bench --skip_wasted_D_wgrad"; timeout -k 10 300 python bench.py --skip_wasted_D_wgrad --no_cpu_baseline > $OUT/bench_skip.json 2>> $OUT/bench.err; echo "exit $?"
echo "== bench n_update_G=1"; timeout -k 10 300 python bench.py --n_update_G 1 --no_cpu_baseline > $OUT/bench_nug1.json 2>> $OUT/bench.err; echo "exit $?"
echo "== bench --workload cgan"; timeout -k 10 500 python bench.py --workload cgan --steps 50 --warmup 5 > $OUT/bench_cgan.json 2>> $OUT/bench.err; echo "exit $?"
echo "== bench --workload twostage_cycle"; timeout -k 10 300 python bench.py --workload twostage_cycle --steps 30 --warmup 5 > $OUT/bench_twostage.json 2>> $OUT/bench.err; echo "exit $?"
cut -c1-300 $OUT/bench.json
