#!/bin/bash
# r3m: why is the product's pass 1 (22.9 us in the trace) slower than the microbenchmark of the same per-block code (15.4 us)?
set -u
TAG=${1:-r3m}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I. -o /tmp/light_path profiles/microbench/light_path.hip 2> "$OUT/build.log" || { cat "$OUT/build.log"; exit 1; }
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_mb" -o t -- /tmp/light_path > "$OUT/light_path_traced.txt" 2>&1)
f=$(find "$OUT/trace_mb" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/microbench_kernel_stats.csv" && cut -d, -f1-6 "$f"; rm -rf "$OUT/trace_mb"
cat "$OUT/light_path_traced.txt" | grep "us per"
b ext_default --workload exterior --no-cpu-baseline --no-extras
b ext_noperiod --workload exterior --no-cpu-baseline --no-extras --opt scan_col_period=0
b ext_w7 --workload exterior --no-cpu-baseline --no-extras --opt scan_waves=7
b ext_w4 --workload exterior --no-cpu-baseline --no-extras --opt scan_waves=4
b ext_noxcd --workload exterior --no-cpu-baseline --no-extras --opt scan_xcd_map=0
trace ext_noperiod --workload exterior --no-extras --opt scan_col_period=0
