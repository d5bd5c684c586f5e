#!/bin/bash
# r3l: light-path microbenchmark (8x8 + XCD order vs 64x1 rows) next to the product's all-exterior tile on the same box
set -u
TAG=${1:-r3l}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I. -o /tmp/light_path profiles/microbench/light_path.hip 2> "$OUT/build.log" || { cat "$OUT/build.log"; exit 1; }
timeout 300 /tmp/light_path 2>&1 | tee "$OUT/light_path.txt"
b exterior --workload exterior --no-cpu-baseline --no-extras
trace exterior --workload exterior --no-extras
