#!/bin/bash
# r3v: what the pre-pass costs per step today: overlapped on the aux stream (default) against serial on the caller's stream
set -u
TAG=${1:-r3v}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
for r in 1 2; do
b cfg2_overlap_$r --no-cpu-baseline --no-extras
b cfg2_serial_$r --no-cpu-baseline --no-extras --opt prepass_overlap=0
b cfg2_noorder_$r --no-cpu-baseline --no-extras --opt order=0
done
trace cfg2_overlap --no-extras --opt cycle_detect=0 --steps 300
trace cfg2_serial --no-extras --opt cycle_detect=0 --opt prepass_overlap=0 --steps 300
