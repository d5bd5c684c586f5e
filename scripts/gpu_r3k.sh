#!/bin/bash
# r3k: fp64 matrix-pipe co-issue microbenchmark (profiles/microbench/mfma_f64_coissue.hip)
set -u
TAG=${1:-r3k}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/mfma_f64 profiles/microbench/mfma_f64_coissue.hip 2> "$OUT/build.log" || { cat "$OUT/build.log"; exit 1; }
timeout 300 /tmp/mfma_f64 2>&1 | tee "$OUT/mfma_f64_coissue.txt"
