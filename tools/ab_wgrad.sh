#!/bin/bash
# A/B of library builds under ab/ on ONE box: per-shape weight-gradient launch times (tools/bench_wgrad.py) per build.
for f in "$@"; do
  echo -n "$f "
  ADX_LIB=$PWD/$f python3 tools/bench_wgrad.py 2>&1 | tail -1
done
