#!/bin/bash
# Round-2 iteration pass: scan-vs-group A/B on the light and headline workloads + kernel traces of the scan passes.
set -u
TAG=${1:-r2b}; WL=${2:-"cfg2 exterior cfg1 chunk_l1 inset"}; KS=${3:-"scan group"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
ab() { W=$1; K=$2; shift 2; TAGX=$(echo "$*" | tr -c 'A-Za-z0-9=\n' '_'); timeout 200 python bench.py --workload $W --kernel $K --no-cpu-baseline "$@" > "$OUT/ab_${W}_${K}${TAGX}.log" 2>&1
  python - "$OUT/ab_${W}_${K}${TAGX}.log" $W $K "$*" <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"{sys.argv[2]:9s} {sys.argv[3]:7s} {sys.argv[4]:24s} occ {r['config'].get('occupancy_api_wg_per_cu')} {r['value']:9.1f} G/s  ms/step {r['ms_per_step']:.4f} kernel_ms avg {r['roofline']['kernel_ms_avg']:.4f} min {r['roofline']['kernel_ms_min']:.4f}  slot_util {r['roofline']['valu_slot_util']:.3f}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e); print(open(sys.argv[1]).read()[-800:])
PY
}
for W in $WL; do for K in $KS; do ab $W $K; done; done
if [ -n "${EXTRA_OPTS:-}" ]; then for W in $WL; do for O in $EXTRA_OPTS; do ab $W scan --opt $O; done; done; fi
trace() { W=$1; K=$2; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_${W}_${K}" -o t -- python "$ROOT/bench.py" --workload $W --kernel $K --no-cpu-baseline --steps 200 --warmup 20 > "$OUT/trace_${W}_${K}.log" 2>&1)
  f=$(find "$OUT/trace_${W}_${K}" -name "*kernel_stats.csv" | head -1); echo "-- $W $K"; [ -n "$f" ] && cut -d, -f1-6 "$f" | head -6; find "$OUT/trace_${W}_${K}" -name "*kernel_trace.csv" -delete; }
for W in ${TRACE_WL:-exterior cfg2}; do trace $W scan; done
if [ "${RUN_TESTS:-1}" = "1" ]; then echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_K:-} > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.log"; fi
du -sh "$OUT"
