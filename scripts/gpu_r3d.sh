#!/bin/bash
# Round-3 fourth GPU pass: knob A/B on the deferred-replay kernels, worker end to end incl. the native sink.
set -u
TAG=${1:-r3d}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
echo "== parity: option matrix"
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "option_matrix" > "$OUT/pytest_matrix.log" 2>&1; echo "pytest(matrix) rc=$?"; tail -2 "$OUT/pytest_matrix.log"
for rep in 1 2; do
b cfg2_base_$rep --no-cpu-baseline --no-extras
b cfg2_exact_long0_$rep --opt exact_long=0 --no-cpu-baseline --no-extras
b cfg2_wpw2_$rep --opt waves_per_wg=2 --no-cpu-baseline --no-extras
done
b cfg2_exact4 --opt exact_steps=4 --no-cpu-baseline --no-extras
b cfg2_exact16 --opt exact_steps=16 --no-cpu-baseline --no-extras
b cfg2_group8 --opt group_steps=8 --no-cpu-baseline --no-extras
b cfg2_order0 --opt order=0 --no-cpu-baseline --no-extras
b cfg3_base --workload cfg3 --no-cpu-baseline --no-extras
b cfg3_exact_long0 --workload cfg3 --opt exact_long=0 --no-cpu-baseline --no-extras
b cfg3_wpw2 --workload cfg3 --opt waves_per_wg=2 --no-cpu-baseline --no-extras
b cfg3_exact0 --workload cfg3 --opt exact_steps=0 --no-cpu-baseline --no-extras
b inset_base --workload inset --no-cpu-baseline --no-extras
b inset_exact_long0 --workload inset --opt exact_long=0 --no-cpu-baseline --no-extras
echo "== worker end to end"
timeout 900 python scripts/worker_e2e.py 12 256 3 > "$OUT/worker_e2e.log" 2>&1; grep -v amdgpu.ids "$OUT/worker_e2e.log" | tail -12
du -sh "$OUT"
