#!/bin/bash
# Build the micro-benchmarks next to their sources (binaries are git-ignored): tools/micro/build.sh
set -euo pipefail
cd "$(dirname "$0")"
for f in *.hip; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -x hip "$f" -o "${f%.hip}.bin"
  echo "built ${f%.hip}.bin"
done
# the same chain with the first 16 dwords of every argument block preloaded into SGPRs
hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 -x hip kernarg_preload.hip -o kernarg_preload_on.bin
echo "built kernarg_preload_on.bin"
