#!/bin/bash
# Quick A/B bench of kernels x workloads on the GPU box.  Usage: scripts/gpu_quick.sh tag "kernels" "workloads" [steps]
set -u
TAG=${1:-q}; KERNELS=${2:-default}; WORKLOADS=${3:-cfg2}; STEPS=${4:-10}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
if [ "${RUN_TESTS:-0}" = "1" ]; then timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.log"; fi
for W in $WORKLOADS; do for K in $KERNELS; do
  timeout 300 python bench.py --steps $STEPS --warmup 2 --kernel $K --workload $W --no-cpu-baseline > "$OUT/bench_${W}_${K}.log" 2>&1
  python - "$OUT/bench_${W}_${K}.log" $W $K <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"{sys.argv[2]:9s} {sys.argv[3]:9s} {r['value']:9.1f} G/s  kernel_ms avg {r['roofline']['kernel_ms_avg']:.4f} min {r['roofline']['kernel_ms_min']:.4f}  slot_util {r['roofline']['valu_slot_util']:.3f}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e); print(open(sys.argv[1]).read()[-1500:])
PY
done; done
