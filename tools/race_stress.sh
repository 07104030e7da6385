#!/bin/bash
# usage (on the GPU box): tools/race_stress.sh     The patch kernel with one wave of every workgroup held back for ~30k cycles after
# the priming barrier (-DSG3P_DELAY_WAVE=1), first WITHOUT the barrier that follows the step-0 fragment reads (the probe must then fail:
# the held-back wave reads step 2's weight tile), then with it (the probe must pass bit for bit).  Both variants are built under /tmp and
# selected through SGAN_HIP_LIB: the product library is never touched (an interrupted run cannot leave the racy build in the tree).
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rc=0
for v in nobarrier barrier; do
  D=/tmp/sgan_race_$v
  rm -rf $D && mkdir -p $D/pkg && cp -r "$ROOT/include" $D/include && cp -r "$ROOT/supervised-gan_amd/csrc" $D/pkg/csrc || exit 2
  X="-DSG3P_DELAY_WAVE=1"; [ $v = nobarrier ] && X="$X -DSG3P_NO_STEP0_BARRIER"
  ( cd $D/pkg/csrc && rm -f *.o libsgan_hip.so && make -j6 EXTRA="$X" > $D/build.log 2>&1 ) || { tail -5 $D/build.log; exit 2; }
  ( cd "$ROOT" && SGAN_HIP_LIB=$D/pkg/csrc/libsgan_hip.so RACE_REPEATS=20 timeout -k 10 300 python tools/race_probe_patch.py 2>&1 | grep -v amdgpu.ids | tail -4 ); r=$?
  echo "== delayed wave, $v: probe exit $r"
  [ $v = nobarrier ] && [ $r -eq 0 ] && { echo "the stress build does not expose the hazard"; rc=1; }
  [ $v = barrier ] && [ $r -ne 0 ] && { echo "FAILED with the barrier in place"; rc=1; }
done
exit $rc
