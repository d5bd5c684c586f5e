#!/bin/bash
# Runs on the MI355X box (via gpurun): smoke, GPU parity tests, VALU microbenchmark, bench, rocprofv3.
# Usage: scripts/gpu_checks.sh [tag]     outputs -> gpurun_out/<tag>/
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12 > "$OUT/rocminfo.txt" 2>&1
nproc > "$OUT/nproc.txt"
echo "== smoke"; timeout 600 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.log"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest_gpu.log"
if [ "${SKIP_MICRO:-0}" != "1" ]; then
  echo "== microbench"; hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates profiles/microbench/valu_rates.hip && timeout 300 /tmp/valu_rates > "$OUT/valu_rates.log" 2>&1; tail -60 "$OUT/valu_rates.log"
fi
echo "== bench"; timeout 600 python bench.py > "$OUT/bench.log" 2>&1; echo "bench rc=$?"; tail -3 "$OUT/bench.log"
for K in ${BENCH_KERNELS:-}; do
  timeout 300 python bench.py --kernel $K --no-cpu-baseline > "$OUT/bench_$K.log" 2>&1; tail -1 "$OUT/bench_$K.log"
done
for W in ${BENCH_WORKLOADS:-}; do
  timeout 300 python bench.py --workload $W --no-cpu-baseline > "$OUT/bench_$W.log" 2>&1; tail -1 "$OUT/bench_$W.log"
done
for W in ${BENCH_F32_WORKLOADS:-}; do
  timeout 600 python bench.py --workload $W --precision f32 --no-cpu-baseline > "$OUT/bench_${W}_f32.log" 2>&1; tail -1 "$OUT/bench_${W}_f32.log"
done
echo "== rocprofv3 kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_trace" -o bench -- python "$ROOT/bench.py" --no-cpu-baseline > "$OUT/prof_trace.log" 2>&1; echo "rocprof rc=$?")
find "$OUT/prof_trace" -name "*stats*" | head; for f in $(find "$OUT/prof_trace" -name "*kernel_stats*.csv" | head -1); do head -8 "$f"; done
if [ "${SKIP_PMC:-0}" != "1" ]; then
  echo "== rocprofv3 pmc (separate pass, no tracing)"
  (cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/prof_pmc" -o bench -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/prof_pmc.log" 2>&1; echo "pmc rc=$?")
  (cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_pmc_w" -o bench -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/prof_pmc_w.log" 2>&1; echo "pmc_w rc=$?")
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_pmc_f" -o bench -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/prof_pmc_f.log" 2>&1; echo "pmc_f rc=$?")
  find "$OUT" -name "*counter_collection*.csv" | head
fi
du -sh "$OUT"
