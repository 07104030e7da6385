#!/bin/bash
# usage (GPU box): tools/abl_thin.sh   -- ablation builds of the thin backward-weight kernel (wrong results, timing only)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/gpurun_out/r3
for a in 0 1 2 3 4 7; do
  D=/tmp/sgan_abl_thin_$a
  rm -rf $D && mkdir -p $D/pkg && cp -r "$ROOT/include" $D/include && cp -r "$ROOT/supervised-gan_amd/csrc" $D/pkg/csrc || exit 2
  ( cd $D/pkg/csrc && rm -f sgan_wgrad.o libsgan_hip.so && make -j6 EXTRA="-DSG_THIN_ABL=$a $ABL_EXTRA" > $D/build.log 2>&1 ) || { tail -5 $D/build.log; exit 2; }
  echo "== SG_THIN_ABL=$a $ABL_EXTRA"
  ( cd "$ROOT" && SGAN_HIP_LIB=$D/pkg/csrc/libsgan_hip.so python tools/bench_thin.py ${WANT:-256:256} 2>&1 | grep "want" )
done
