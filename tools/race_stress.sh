#!/bin/bash
# usage (on the GPU box): tools/race_stress.sh     The patch kernel with one wave of every workgroup held back for ~30k cycles after
# the priming barrier (-DSG3P_DELAY_WAVE=1), first WITHOUT the barrier that follows the step-0 fragment reads (the probe must then fail:
# the held-back wave reads step 2's weight tile), then with it (the probe must pass bit for bit).  Restores the product library.
set -o pipefail
cd supervised-gan_amd/csrc
cp libsgan_hip.so /tmp/lib_keep.so
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DSG3P_DELAY_WAVE=1"
OBJS="sgan_igemm.o sgan_wgrad.o sgan_wgrad3.o sgan_ew.o sgan_fused.o"
rc=0
for v in nobarrier barrier; do
  X=""; [ $v = nobarrier ] && X="-DSG3P_NO_STEP0_BARRIER"
  /opt/rocm/bin/hipcc $FL $X -c sgan_igemm3.hip -o /tmp/ig3_race.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/ig3_race.o -o libsgan_hip.so || { rc=2; break; }
  ( cd ../.. && RACE_REPEATS=20 timeout -k 10 300 python tools/race_probe_patch.py 2>&1 | grep -v amdgpu.ids | tail -4 ); r=$?
  echo "== delayed wave, $v: probe exit $r"
  [ $v = nobarrier ] && [ $r -eq 0 ] && { echo "the stress build does not expose the hazard"; rc=1; }
  [ $v = barrier ] && [ $r -ne 0 ] && { echo "FAILED with the barrier in place"; rc=1; }
done
cp /tmp/lib_keep.so libsgan_hip.so
exit $rc
