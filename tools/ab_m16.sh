#!/bin/bash
# What the 16x16x32 MFMA shape would buy per kernel family (timing-only build libadx_m16.so: csrc/build.sh -DADX_HS_M16_TIMING),
# same box, alternating runs: weight gradients, fp32-layout convs (training forward / data gradient, stride 2, stem), training step
P=$PWD/autonomous_driving_with_diffusion_model_amd
for rnd in 1 2; do
  for lib in libadx.so libadx_m16.so; do
    echo "== $rnd $lib"
    ADX_LIB=$P/$lib RANGE=0 python3 tools/bench_wgrad.py 2>&1 | tail -1
    ADX_LIB=$P/$lib python3 tools/bench_conv.py 2>&1 | grep -- "->"
  done
done
python3 tools/ab_train.py libadx.so libadx_m16.so
