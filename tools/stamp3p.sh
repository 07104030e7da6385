#!/bin/bash
# usage (on the GPU box): tools/stamp3p.sh [extra -D flags]    stamps of sg_igemm3p_kernel on the D3 layer (6 and 3 problems)
set -o pipefail
cd supervised-gan_amd/csrc
cp libsgan_hip.so /tmp/lib_keep.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DSG3P_STAMP "$@" -c sgan_igemm3.hip -o /tmp/ig3_st.o || exit 1
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 sgan_igemm.o /tmp/ig3_st.o sgan_wgrad.o sgan_wgrad3.o sgan_ew.o sgan_fused.o -o libsgan_hip.so
cd ../..
if [ -n "$STAMP_G" ]; then timeout -k 10 120 python tools/stamp3p.py gfwd 1 2>&1 | grep -v amdgpu.ids; else for op in fwd dgrad; do for n in 6 3; do timeout -k 10 120 python tools/stamp3p.py $op $n 2>&1 | grep -v amdgpu.ids; done; done; fi
cp /tmp/lib_keep.so supervised-gan_amd/csrc/libsgan_hip.so
