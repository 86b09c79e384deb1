#!/bin/bash
# The reference-faithful step (tools/faithful_only.py, 100 timed steps) over several builds of libadx on ONE GPU, alternating.
# Usage: bash tools/ab_faithful.sh lib1.so lib2.so ...   (paths relative to the repo root)
for r in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib: "
    ADX_LIB=$PWD/$lib STEPS=103 python tools/faithful_only.py 2>/dev/null | tail -1
  done
done
