#!/bin/bash
# A/B of per-shape cell-launch times (tools/cells_modes.py) between builds of libadx on ONE box: tools/ab_cells.sh libadx.so libadx_x.so
for rnd in 1 2; do
  for lib in "$@"; do
    echo -n "$rnd $lib "
    ADX_LIB=$PWD/autonomous_driving_with_diffusion_model_amd/$lib python tools/cells_modes.py 2>&1 | tail -1
  done
done
