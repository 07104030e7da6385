#!/bin/bash
# usage (on the GPU box): tools/race_stress_all.sh [wave ...]     (default: 0 3)
# Builds a second copy of the library with -DSG_STRESS_DELAY=<wave> (sgan_common.h: that wave of every workgroup sleeps ~6400 cycles
# after every barrier; waves 4-7 exist in the 512-thread kernels only) under /tmp and runs the whole GPU test suite on it through SGAN_HIP_LIB.  A missing barrier between the last
# read of an LDS buffer and the store that recycles it then fails the parity tests every time.  The product library is not touched.
# Sensitivity check: STRESS_EXTRA=-DSG3P_NO_STEP0_BARRIER tools/race_stress_all.sh 1   must FAIL (the hazard fixed in round 2).
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rc=0
for w in ${@:-0 3}; do
  D=/tmp/sgan_stress_$w
  rm -rf $D && mkdir -p $D/pkg && cp -r "$ROOT/include" $D/include && cp -r "$ROOT/supervised-gan_amd/csrc" $D/pkg/csrc || exit 2
  ( cd $D/pkg/csrc && rm -f *.o libsgan_hip.so && make -j6 EXTRA="-DSG_STRESS_DELAY=$w $STRESS_EXTRA" > $D/build.log 2>&1 ) || { tail -5 $D/build.log; exit 2; }
  echo "== built with wave $w delayed: $(date +%T)"
  ( cd "$ROOT" && SGAN_HIP_LIB=$D/pkg/csrc/libsgan_hip.so timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 ); r=$?
  echo "== wave $w delayed: pytest exit $r"
  [ $r -ne 0 ] && rc=1
done
exit $rc
