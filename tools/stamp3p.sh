#!/bin/bash
# usage (on the GPU box): tools/stamp3p.sh [extra -D flags]    stamps of sg_igemm3p_kernel (built under /tmp, selected through SGAN_HIP_LIB)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=/tmp/sgan_stamp3p
rm -rf $D && mkdir -p $D/pkg && cp -r "$ROOT/include" $D/include && cp -r "$ROOT/supervised-gan_amd/csrc" $D/pkg/csrc || exit 2
( cd $D/pkg/csrc && rm -f sgan_igemm3.o libsgan_hip.so && make -j6 EXTRA="-DSG3P_STAMP $*" sgan_igemm3.o > $D/build.log 2>&1 && make -j6 >> $D/build.log 2>&1 ) || { tail -5 $D/build.log; exit 2; }
cd "$ROOT"
export SGAN_HIP_LIB=$D/pkg/csrc/libsgan_hip.so
for spec in ${STAMP_OPS:-"fwd 6" "fwd 3" "s2fwd1 6" "s2fwd2 6" "gfwd 1"}; do timeout -k 10 120 python tools/stamp3p.py $spec 2>&1 | grep -v amdgpu.ids; done
