#!/bin/bash
# r3o: statistics fused into the finish-in-place light pass (DataChunks of all-exterior tiles write bytes only)
set -u
TAG=${1:-r3o}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_serializer.py tests/test_server_storage.py -x -q -m gpu -k "scan_finishes or (option_matrix and scan) or golden or seeded_views or ragged or mrd_edge or f32_variant or history_independent or any_arrival or many_streams or lazy_uniform or two_tiles or worker or serial or stats_reduction or slot0 or cfg3_as or farm or native or end_to_end" > "$OUT/pytest_focus.log" 2>&1; echo "pytest(focus) rc=$?"; tail -5 "$OUT/pytest_focus.log"
timeout 300 python scripts/level_rate.py > "$OUT/level16.log" 2>&1; tail -4 "$OUT/level16.log"
b ext_default --workload exterior --no-cpu-baseline --no-extras
b cfg2_full --no-cpu-baseline
