#!/bin/bash
# r3t: cycle-aware dispatch-order probe (MBK_OPT_PROBE_CYCLE): the cycle-test leg with 0 / 64 / 128 / 256
set -u
TAG=${1:-r3t}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "option_matrix and probe_cycle" > "$OUT/pytest_focus.log" 2>&1; echo "pytest(focus) rc=$?"; tail -3 "$OUT/pytest_focus.log"
for d in 0 64 128 256 0 128; do
b cfg2_pc$d --no-cpu-baseline --no-extras --opt cycle_detect=1 --opt probe_cycle=$d
done
for d in 0 128; do
b l1_pc$d --workload chunk_l1 --no-cpu-baseline --no-extras --opt cycle_detect=1 --opt probe_cycle=$d
b cfg3_pc$d --workload cfg3 --no-cpu-baseline --no-extras --opt cycle_detect=1 --opt probe_cycle=$d
b cfg5_pc$d --workload cfg5 --no-cpu-baseline --no-extras --opt cycle_detect=1 --opt probe_cycle=$d
timeout 300 python scripts/level_rate.py 16 1024 probe_cycle=$d 2>&1 | grep -v amdgpu | tail -3
done
