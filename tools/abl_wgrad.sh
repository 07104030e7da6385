#!/bin/bash
# usage (on the GPU box): tools/abl_wgrad.sh  -> backward-weight launches with and without their gradient atomics (diagnostics build)
cd supervised-gan_amd/csrc
cp libsgan_hip.so /tmp/lib_keep.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DSGW3_NO_ATOMICS -c sgan_wgrad3.hip -o /tmp/wg3.o || exit 1
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 sgan_igemm.o sgan_igemm3.o sgan_wgrad.o /tmp/wg3.o sgan_ew.o -o libsgan_hip.so
cd ../..
echo "== no atomics"; timeout -k 10 200 python tools/bench_igemm3.py wgrad --modes bf16x3 --tiles auto --only D1x6,D2x6,D3x6,D3x3,D2x3,D1x3,G1,G2,G3,G4 2>&1 | grep -v "amdgpu\|launch op" | awk '{print $1, $5, $6}' | tr "\n" ";"; echo
cp /tmp/lib_keep.so supervised-gan_amd/csrc/libsgan_hip.so
echo "== normal"; timeout -k 10 200 python tools/bench_igemm3.py wgrad --modes bf16x3 --tiles auto --only D1x6,D2x6,D3x6,D3x3,D2x3,D1x3,G1,G2,G3,G4 2>&1 | grep -v "amdgpu\|launch op" | awk '{print $1, $5, $6}' | tr "\n" ";"; echo
