#!/bin/bash
# r3x: how much of the heavy-first order's value hangs on the probe depth (is "inside after 4 steps" -- scan's criterion -- enough)?
set -u
TAG=${1:-r3x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
for d in 32 4 8 16 64 128 32; do
b cfg2_probe$d --kernel group --no-cpu-baseline --no-extras --opt probe_steps=$d
done
