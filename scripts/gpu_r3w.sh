#!/bin/bash
# r3w: BASELINE cfg3 in its DataChunk form through the N > 1 queue (2 x 2 tiles of 4096^2 = the 8192^2 image), one rank and two ranks sharing the GPU
set -u
TAG=${1:-r3w}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
b cfg3_one --workload cfg3 --no-cpu-baseline --no-extras --steps 6
b cfg3_queue_n1 --shard queue --workload cfg3 --grid 2 --steps 6 --no-cpu-baseline
b cfg3_queue_n2_oversub --gpus 2 --oversubscribe --workload cfg3 --grid 2 --steps 6
b cfg3_queue_g4_n1 --shard queue --workload cfg3 --grid 4 --steps 2 --no-cpu-baseline
