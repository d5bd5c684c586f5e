#!/bin/bash
# Round-3 first GPU pass: tests, the headline with its new extra objects, the dispatch-order A/B, the N>1 path
# exercised for real on one GPU (two ranks sharing GPU 0), queue mode at N=1.  Usage: scripts/gpu_r3a.sh TAG
set -u
TAG=${1:-r3a}; SKIP=" ${SKIP:-} "
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
skip() { [[ "$SKIP" == *" $1 "* ]]; }
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12 > "$OUT/rocminfo.txt" 2>&1; nproc > "$OUT/nproc.txt"
echo "== smoke"; timeout 600 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
line() { python - "$1" <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); ro=r["roofline"]; cy=r.get("cycle_detection")
    print(f"  {r['config']['workload'][:9]:9s} {r['config']['kernel']:8s} {r['dtype']} n={r['n_gpus']} {r['config']['shard']:5s} {str(r['config'].get('options')):22s} {r['value']:9.1f} G/s  ms/step {r['ms_per_step']:.4f} launch ms avg {ro['kernel_ms_avg']:.4f} med {ro.get('kernel_ms_median',0):.4f} min {ro['kernel_ms_min']:.4f}  frac {ro['frac']:.3f} slot_util {(ro['valu_slot_util'] or 0):.3f}"
          + (f" | cycle test on: {cy['value']:.1f} G/s-eq {cy['ms_per_step']:.4f} ms x{cy['speedup_vs_strict']:.2f} same={cy['same_pixel_iterations_and_never_count']}" if cy else ""))
    for k in ("end_to_end", "queue_job"):
        if k in r: print("    ", k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in r[k].items() if a != "what"})
    c=r["config"]
    if c["shard"] != "own": print("     once", c.get("tiles_exactly_once", c.get("bands_exactly_once")), "per rank", c.get("tiles_per_rank", c.get("bands_per_rank")), "finish ms", c["rank_finish_ms"], "gpus", c["distinct_gpus"], [x["pci_bus_id"] for x in c["ranks_seen"]])
except Exception as e:
    print("  FAILED", sys.argv[1], e); print(open(sys.argv[1]).read()[-800:])
PY
}
b() { name=$1; shift; timeout 900 python bench.py "$@" > "$OUT/bench_$name.log" 2>&1; line "$OUT/bench_$name.log"; }
echo "== bench (headline first, with the CPU baseline and the extra objects)"
b cfg2_default
b cfg2_two_class --opt probe_mid=65537 --no-cpu-baseline --no-extras
b cfg2_default_b --no-cpu-baseline --no-extras
b cfg2_two_class_b --opt probe_mid=65537 --no-cpu-baseline --no-extras
b cfg2_mid3 --opt probe_mid=3 --no-cpu-baseline --no-extras
b cfg2_mid12 --opt probe_mid=12 --no-cpu-baseline --no-extras
b cfg2_scan --kernel scan --no-cpu-baseline --no-extras
echo "== queue mode / N>1 functional (two ranks on one GPU)"
b queue_n1 --shard queue --no-cpu-baseline
b queue_n2_oversub --gpus 2 --oversubscribe --no-cpu-baseline
b own_n2_oversub --gpus 2 --oversubscribe --shard own --no-cpu-baseline --steps 100
b bands_n2_oversub --gpus 2 --oversubscribe --shard bands --workload cfg3 --no-cpu-baseline --steps 4
echo "== other workloads"
b exterior --workload exterior --no-cpu-baseline; b exterior_both --workload exterior --outputs both --no-cpu-baseline
b chunk_l1 --workload chunk_l1 --no-cpu-baseline; b inset --workload inset --no-cpu-baseline
b cfg1 --workload cfg1 --no-cpu-baseline
b cfg3 --workload cfg3 --no-cpu-baseline
if ! skip tests; then echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"; grep -E "^FAILED|^ERROR" "$OUT/pytest_gpu.log" | cut -c1-220 | head -10; fi
if ! skip e2e; then echo "== level rate"; timeout 200 python scripts/level_rate.py 16 1024 > "$OUT/level16.log" 2>&1; grep "level\|two" "$OUT/level16.log"; fi
du -sh "$OUT"
