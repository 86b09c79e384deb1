#!/bin/bash
# A/B of whole UNet forwards (tools/chain_time.py: graphs of 20 forwards at 128 rows / H 32 and 2 rows / H 16) over several builds of
# libadx on ONE GPU, alternating.  Usage: bash tools/ab_chain_time.sh lib1.so lib2.so ...
for rnd in 0 1; do
  for lib in "$@"; do
    ADX_LIB=$PWD/$lib TAG=$(basename $lib) python tools/chain_time.py 2>&1 | tail -1
  done
done
