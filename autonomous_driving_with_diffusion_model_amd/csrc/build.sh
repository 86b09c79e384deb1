#!/bin/bash
# Build libadx.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [extra hipcc flags]
set -euo pipefail
cd "$(dirname "$0")"
OUT=${ADX_OUT:-../libadx.so}
SRCS="api.cpp batch_ops.hip tconv.hip embed.hip sched.hip unet.hip conv2d.hip conv2d_hs.hip conv2d_wgrad_hs.hip trajpred.hip tbwd.hip unet_train.hip resnet_train.hip optim.hip"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -x hip $SRCS -o "$OUT" \
  -Wall -Wno-unused-function "$@"
echo "built $(readlink -f $OUT)"
