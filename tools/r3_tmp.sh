mkdir -p gpurun_out/r3
python tools/bench_thin.py 256:256 128:256 2>&1 | grep want
for v in 0 1; do echo "== HIP_FORCE_DEV_KERNARG=$v"; HIP_FORCE_DEV_KERNARG=$v python bench.py --no_cpu_baseline --no_kernel_profile --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'])"; done
for v in 0 1; do echo "== HIP_FORCE_DEV_KERNARG=$v"; HIP_FORCE_DEV_KERNARG=$v python bench.py --no_cpu_baseline --no_kernel_profile --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'])"; done
