#!/bin/bash
# streams in flight for the cursor modes
set -u
TAG=${1:-r3g}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
b cfg3_one --workload cfg3 --no-cpu-baseline --no-extras --steps 6
for S in 1 2 3 4 6 8; do b bands_s$S --shard bands --workload cfg3 --steps 6 --streams $S --no-cpu-baseline; done
for R in 256 1024 2048; do b bands_r${R}_s4 --shard bands --workload cfg3 --steps 6 --streams 4 --band-rows $R --no-cpu-baseline; done
b bands_r128_s8 --shard bands --workload cfg3 --steps 6 --streams 8 --band-rows 128 --no-cpu-baseline
b bands_r128_s16 --shard bands --workload cfg3 --steps 6 --streams 16 --band-rows 128 --no-cpu-baseline
for S in 1 2 3 4; do b queue_s$S --shard queue --streams $S --no-cpu-baseline; done
b queue_n2_s2 --gpus 2 --oversubscribe --streams 2
b queue_n2_s4 --gpus 2 --oversubscribe --streams 4
du -sh "$OUT"
