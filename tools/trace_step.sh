#!/bin/bash
# usage (GPU box): tools/trace_step.sh <tag> [bench args]  -> gpurun_out/r3/<tag>_timeline.txt: one replayed step, kernel by kernel
T=$1; shift
mkdir -p gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r3/trace_$T
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3/trace_$T -- python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_kernel_profile "$@" > gpurun_out/r3/trace_$T.log 2>&1; echo "rocprof exit $?"
f=$(ls gpurun_out/r3/trace_$T/*/*_kernel_trace.csv | tail -1)
python tools/step_timeline.py $f --full > gpurun_out/r3/${T}_timeline.txt
tail -40 gpurun_out/r3/${T}_timeline.txt
rm -rf gpurun_out/r3/trace_$T
