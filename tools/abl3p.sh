#!/bin/bash
# usage (on the GPU box): tools/abl3p.sh "<bench_igemm3 args>" abl1 abl2 ...   -> per-ablation timing of sg_igemm3p_kernel (diagnostics builds)
set -o pipefail
ARGS=$1; shift
cd supervised-gan_amd/csrc
cp libsgan_hip.so /tmp/lib_keep.so
for a in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DSG3P_ABL=$a -c sgan_igemm3.hip -o /tmp/ig3_abl.o 2>/dev/null || { echo "build $a failed"; continue; }
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 sgan_igemm.o /tmp/ig3_abl.o sgan_wgrad.o sgan_wgrad3.o sgan_ew.o -o libsgan_hip.so
  echo "== SG3P_ABL=$a"
  (cd ../.. && timeout -k 10 200 python tools/bench_igemm3.py $ARGS 2>&1 | grep -v amdgpu.ids | tail -n +2)
done
cp /tmp/lib_keep.so libsgan_hip.so
