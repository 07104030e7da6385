#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=/tmp/sgan_stamp_thin
rm -rf $D && mkdir -p $D/pkg && cp -r "$ROOT/include" $D/include && cp -r "$ROOT/supervised-gan_amd/csrc" $D/pkg/csrc || exit 2
( cd $D/pkg/csrc && rm -f sgan_wgrad.o libsgan_hip.so && make -j6 EXTRA="-DSGTHIN_STAMP $ABL_EXTRA" > $D/build.log 2>&1 ) || { tail -5 $D/build.log; exit 2; }
cd "$ROOT" && PYTHONPATH=tools SGAN_HIP_LIB=$D/pkg/csrc/libsgan_hip.so python tools/stamp_thin.py 2>&1 | grep -v amdgpu.ids
