#!/bin/bash
# usage (GPU box): tools/abl_head.sh   -- tile-height and ablation builds of the head backward kernel (ablations: wrong results, timing only)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/gpurun_out/r3
( cd "$ROOT" && PROBE_GENERIC=1 python tools/probe_head.py 2>&1 | grep "^head" )
for v in ${VARIANTS:-4:0 8:8 8:4 16:0}; do
  ty=${v%%:*}; a=${v##*:}
  D=/tmp/sgan_abl_head_${ty}_$a
  rm -rf $D && mkdir -p $D/pkg && cp -r "$ROOT/include" $D/include && cp -r "$ROOT/supervised-gan_amd/csrc" $D/pkg/csrc || exit 2
  ( cd $D/pkg/csrc && rm -f sgan_head.o libsgan_hip.so && make -j6 EXTRA="-DSGH_TY=$ty -DSGH_ABL=$a" > $D/build.log 2>&1 ) || { tail -5 $D/build.log; exit 2; }
  echo "== SGH_TY=$ty SGH_ABL=$a"
  ( cd "$ROOT" && SGAN_HIP_LIB=$D/pkg/csrc/libsgan_hip.so python tools/probe_head.py 2>&1 | grep "^head" )
done
