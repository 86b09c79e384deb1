#!/bin/bash
# Build libadx.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [extra hipcc flags]
#
# One object per source, compiled in parallel and cached under csrc/build/ by content hash (source + every header +
# flags), then linked.  The library carries the sha256 of ALL its sources (adx_source_hash()), which is how
# __graft_entry__.build() decides whether the .so on disk was produced from the sources on disk.
set -euo pipefail
cd "$(dirname "$0")"
OUT=${ADX_OUT:-../libadx.so}
SRCS="api.cpp batch_ops.hip tconv.hip tconv_hs.hip tconv_generic.hip tconv_pack.hip tconv_chain.hip tconv_pipe.hip embed.hip sched.hip augment.hip unet.hip conv2d.hip conv2d_hs.hip conv2d_hs16.hip conv2d_wgrad_hs.hip conv2d_wgrad_stem_hs.hip trajpred.hip tbwd.hip unet_train.hip resnet_train.hip optim.hip probe.hip"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function $*"
HDRS=$(ls *.h ../../include/adx.h | LC_ALL=C sort)
SRC_HASH=$(cat $(ls *.hip *.cpp *.h ../../include/adx.h build.sh | LC_ALL=C sort) | sha256sum | cut -c1-32)
HDR_HASH=$(cat $HDRS | sha256sum | cut -c1-16)
OBJDIR=${ADX_OBJDIR:-build}
mkdir -p "$OBJDIR"
JOBS=${ADX_JOBS:-$(nproc)}

compile_one() {
  local src=$1 extra=""
  [ "$src" = api.cpp ] && extra="-DADX_SRC_HASH=\"$SRC_HASH\""
  local key
  key=$( (cat "$src"; echo "$HDR_HASH $FLAGS $extra") | sha256sum | cut -c1-16)
  local obj="$OBJDIR/${src%.*}.$key.o"
  if [ ! -f "$obj" ]; then
    rm -f "$OBJDIR/${src%.*}".*.o
    hipcc $FLAGS $extra -x hip -c "$src" -o "$obj.tmp" && mv "$obj.tmp" "$obj"
  fi
}
export -f compile_one
export FLAGS HDR_HASH SRC_HASH OBJDIR
printf '%s\n' $SRCS | xargs -P "$JOBS" -I{} bash -c 'compile_one {}'
OBJS=""
for s in $SRCS; do OBJS="$OBJS $(ls $OBJDIR/${s%.*}.*.o)"; done
hipcc --offload-arch=gfx950 -fPIC -shared $OBJS -o "$OUT.tmp"
mv "$OUT.tmp" "$OUT"
echo "built $(readlink -f $OUT) (sources $SRC_HASH)"
