#!/bin/bash
# Reference-faithful step time by the number of sub-batch streams of the perception pass (ADX_RESNET_STREAMS), same box, alternating
for rnd in 1 2; do
  for n in 1 2 3 4; do
    echo -n "$rnd streams=$n "
    ADX_RESNET_STREAMS=$n python tools/faithful_only.py 2>&1 | tail -1
  done
done
