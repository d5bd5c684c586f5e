#!/bin/bash
# r3r: the all-exterior DataChunk as the worker asks for it (bytes + statistics): fused statistics against the counts pass
set -u
TAG=${1:-r3r}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
for v in "fused" "twopass scan_inline=0"; do set -- $v; name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$name" -o t -- python "$ROOT/scripts/light_chunk_rate.py" "$@" > "$OUT/light_chunk_$name.txt" 2>&1)
  grep DataChunk "$OUT/light_chunk_$name.txt"
  f=$(find "$OUT/trace_$name" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/light_chunk_${name}_kernel_stats.csv" && cut -d, -f1-4 "$f" | head -6; rm -rf "$OUT/trace_$name"
done
