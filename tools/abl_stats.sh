#!/bin/bash
# usage (on the GPU box): tools/abl_stats.sh   -> bench with and without the global statistics atomics (diagnostics build, wrong results)
set -o pipefail
cd supervised-gan_amd/csrc
cp libsgan_hip.so /tmp/lib_keep.so
for f in sgan_igemm sgan_igemm3; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DSG_NO_STAT_ATOMICS -c $f.hip -o /tmp/$f.o || exit 1; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/sgan_igemm.o /tmp/sgan_igemm3.o sgan_wgrad.o sgan_wgrad3.o sgan_ew.o -o libsgan_hip.so
cd ../..
echo "== without the statistics atomics"; timeout -k 10 200 python bench.py --no_cpu_baseline --no_kernel_profile 2>/dev/null | cut -c1-250
cp /tmp/lib_keep.so supervised-gan_amd/csrc/libsgan_hip.so
echo "== normal build"; timeout -k 10 200 python bench.py --no_cpu_baseline --no_kernel_profile 2>/dev/null | cut -c1-250
