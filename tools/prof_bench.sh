#!/bin/bash
# usage (GPU box): tools/prof_bench.sh <tag> [env ...]   rocprofv3 kernel trace + stats of the graph-replayed bench -> gpurun_out/r3/<tag>_stats.txt
T=$1; shift
mkdir -p gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for e in "$@"; do export "$e"; done
rm -rf gpurun_out/r3/prof_$T
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/prof_$T -- python bench.py --steps 100 --warmup 10 --no_cpu_baseline --no_kernel_profile > gpurun_out/r3/prof_$T.log 2>&1; echo "rocprof exit $?"
f=$(ls gpurun_out/r3/prof_$T/*/*_kernel_stats.csv | tail -1)
cp $f gpurun_out/r3/${T}_kernel_stats.csv
python tools/step_kernels.py $f 112 > gpurun_out/r3/${T}_stats.txt
tail -1 gpurun_out/r3/${T}_stats.txt; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r3/prof_$T.log
rm -rf gpurun_out/r3/prof_$T
