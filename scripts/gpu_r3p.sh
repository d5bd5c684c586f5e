#!/bin/bash
# r3p: where does kernel "scan" lose to "group" on cfg2?  kernel traces of both
set -u
TAG=${1:-r3p}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
trace cfg2_scan --kernel scan --no-extras --opt cycle_detect=0 --steps 200
trace cfg2_group --kernel group --no-extras --opt cycle_detect=0 --steps 200
trace cfg2_scan_np --kernel scan --no-extras --opt cycle_detect=0 --opt scan_col_period=0 --steps 200
b cfg2_scan_e0 --kernel scan --no-cpu-baseline --no-extras --opt exact_steps=0
b cfg2_scan_p1 --kernel scan --no-cpu-baseline --no-extras --opt scan_col_period=1
b cfg2_scan_p16 --kernel scan --no-cpu-baseline --no-extras --opt scan_col_period=16
