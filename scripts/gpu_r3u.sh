#!/bin/bash
# r3u: longer randomized soak on the final code (two more seeds; antenna windows added to the view kinds)
set -u
TAG=${1:-r3u}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
timeout 400 python scripts/gpu_soak.py 150 29 > "$OUT/soak_29.log" 2>&1; tail -2 "$OUT/soak_29.log"
timeout 400 python scripts/gpu_soak.py 150 41 > "$OUT/soak_41.log" 2>&1; tail -2 "$OUT/soak_41.log"
