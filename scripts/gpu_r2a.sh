#!/bin/bash
# Round-2 first GPU pass: probe the new scan kernel, full GPU test suite, kernel A/B, microbench, power traces.
set -u
TAG=${1:-r2a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
echo "== probe scan"; timeout 180 python scripts/gpu_probe.py scan > "$OUT/probe_scan.log" 2>&1; echo "probe rc=$?"; tail -8 "$OUT/probe_scan.log"
echo "== A/B"
ab() { W=$1; K=$2; shift 2; timeout 200 python bench.py --workload $W --kernel $K --no-cpu-baseline "$@" > "$OUT/ab_${W}_${K}.log" 2>&1
  python - "$OUT/ab_${W}_${K}.log" $W $K <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"{sys.argv[2]:9s} {sys.argv[3]:7s} {r['value']:9.1f} G/s  ms/step {r['ms_per_step']:.4f} kernel_ms avg {r['roofline']['kernel_ms_avg']:.4f} min {r['roofline']['kernel_ms_min']:.4f}  slot_util {r['roofline']['valu_slot_util']:.3f}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e); print(open(sys.argv[1]).read()[-800:])
PY
}
for W in cfg2 exterior cfg1 chunk_l1 inset; do for K in scan group; do ab $W $K; done; done
for K in scan group refill; do ab cfg3 $K; done
ab cfg4 scan; ab cfg4 group
ab cfg5 scan; ab cfg5 group
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -6 "$OUT/pytest_gpu.log"
echo "== microbench (round-2 kinds)"; hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates profiles/microbench/valu_rates.hip && timeout 300 /tmp/valu_rates new > "$OUT/valu_rates_r2.log" 2>&1; grep "waves/SIMD=8\|waves/SIMD=4" "$OUT/valu_rates_r2.log"
echo "== power traces"
for spec in "cfg3 group" "cfg3 refill" "cfg3 scan" "inset scan" "cfg2 scan" "exterior scan"; do set -- $spec
  timeout 200 python scripts/power_trace.py "$OUT/power_$1_$2.json" -- python bench.py --workload $1 --kernel $2 --no-cpu-baseline --steps $([ $1 = cfg3 ] && echo 150 || ([ $1 = inset ] && echo 600 || echo 4000)) > "$OUT/power_$1_$2.log" 2>&1
  grep "power_trace" "$OUT/power_$1_$2.log" | cut -c1-900
done
echo "== level rate"; timeout 200 python scripts/level_rate.py 16 1024 > "$OUT/level16.log" 2>&1; tail -3 "$OUT/level16.log"
du -sh "$OUT"
