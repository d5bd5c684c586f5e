#!/bin/bash
set -u
TAG=${1:-r3h}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
b cfg3_one --workload cfg3 --no-cpu-baseline --no-extras --steps 6
b bands_n1 --shard bands --workload cfg3 --steps 6 --no-cpu-baseline
b bands_n1_r128 --shard bands --workload cfg3 --steps 6 --band-rows 128 --no-cpu-baseline
b bands_n2 --gpus 2 --oversubscribe --shard bands --workload cfg3 --steps 6
b bands_n2_r128 --gpus 2 --oversubscribe --shard bands --workload cfg3 --steps 6 --band-rows 128
b queue_n1 --shard queue --no-cpu-baseline
b queue_n2 --gpus 2 --oversubscribe
