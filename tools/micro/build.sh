#!/bin/bash
# Build the micro-benchmarks next to their sources (binaries are git-ignored): tools/micro/build.sh
set -euo pipefail
cd "$(dirname "$0")"
for f in *.hip; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -x hip "$f" -o "${f%.hip}.bin"
  echo "built ${f%.hip}.bin"
done
