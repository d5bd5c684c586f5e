#!/bin/bash
# r3q: 32-step groups for the strict fp64 loop (A/B against 16), parity of the new option values
set -u
TAG=${1:-r3q}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "option_matrix and group_steps" > "$OUT/pytest_focus.log" 2>&1; echo "pytest(focus) rc=$?"; tail -3 "$OUT/pytest_focus.log"
for r in 1 2; do
b cfg2_g16_$r --no-cpu-baseline --no-extras
b cfg2_g32_$r --no-cpu-baseline --no-extras --opt group_steps=32
done
b inset_g16 --workload inset --no-cpu-baseline --no-extras
b inset_g32 --workload inset --no-cpu-baseline --no-extras --opt group_steps=32
b cfg3_g16 --workload cfg3 --no-cpu-baseline --no-extras
b cfg3_g32 --workload cfg3 --no-cpu-baseline --no-extras --opt group_steps=32
