#!/bin/bash
# One command under the values of one environment switch, alternating, on ONE GPU.
# Usage: bash tools/ab_env.sh VAR "v1 v2 ..." ROUNDS -- command ...
var=$1; vals=$2; rounds=$3; shift 4
for r in $(seq 1 $rounds); do
  for v in $vals; do
    echo "== $var=$v (round $r)"
    env $var=$v "$@" 2>&1 | tail -3
  done
done
