#!/bin/bash
# r3n: light pass as one asm loop per run of blocks (exact EXEC narrowing, no ring check) + finish-in-place form without pass 2
set -u
TAG=${1:-r3n}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scan_finishes or (option_matrix and scan) or golden_full_datachunks or seeded_views or ragged or mrd_edge or cycle_detection_is_bit_exact or f32_variant or history_independent or any_arrival or many_streams or lazy_uniform" > "$OUT/pytest_focus.log" 2>&1; echo "pytest(focus) rc=$?"; tail -5 "$OUT/pytest_focus.log"
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I. -o /tmp/light_path profiles/microbench/light_path.hip 2> "$OUT/build.log" || cat "$OUT/build.log"
timeout 300 /tmp/light_path 2>&1 | tee "$OUT/light_path.txt"
b ext_default --workload exterior --no-cpu-baseline --no-extras
b ext_twopass --workload exterior --no-cpu-baseline --no-extras --opt scan_inline=0
b ext_twopass_np --workload exterior --no-cpu-baseline --no-extras --opt scan_inline=0 --opt scan_col_period=0
b ext_both --workload exterior --no-cpu-baseline --no-extras --outputs both
b cfg2_scan --kernel scan --no-cpu-baseline --no-extras
b cfg2_default --no-cpu-baseline --no-extras
trace ext_default --workload exterior --no-extras
trace ext_twopass --workload exterior --no-extras --opt scan_inline=0
timeout 300 python scripts/level_rate.py > "$OUT/level16.log" 2>&1; tail -8 "$OUT/level16.log"
