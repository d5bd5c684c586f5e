#!/bin/bash
# Round-2 follow-up pass: cycle test + interior fast path.  Tests first, then the headline bench (strict leg +
# cycle leg), kernel traces and one PMC pass for each setting, a few other workloads, the level rate.
set -u
TAG=${1:-r2w}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
T0=$(date +%s); stamp() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12 > "$OUT/rocminfo.txt" 2>&1; nproc > "$OUT/nproc.txt"
stamp smoke; timeout 400 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
if [ "${RUN_TESTS:-1}" = 1 ]; then stamp pytest; timeout 420 python -m pytest tests -m gpu -q --maxfail=6 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.log"; grep -E "^FAILED|^ERROR" "$OUT/pytest_gpu.log" | cut -c1-220 | head -12; fi
line() { python - "$1" <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); ro=r["roofline"]; cy=r.get("cycle_detection")
    print(f"  {r['config']['workload'][:9]:9s} {r['config']['kernel']:8s} {r['dtype']} {str(r['config'].get('options')):22s} {r['value']:9.1f} G/s  launch ms avg {ro['kernel_ms_avg']:.4f} min {ro['kernel_ms_min']:.4f}  frac {ro['frac']:.3f} util {(ro['valu_slot_util'] or 0):.3f}"
          + (f"  | cycle on: {cy['value']:.1f} G/s-eq {cy['ms_per_step']:.4f} ms x{cy['speedup_vs_strict']:.2f} same={cy['same_pixel_iterations_and_never_count']}" if cy else ""))
except Exception as e:
    print("  FAILED", sys.argv[1], e); print(open(sys.argv[1]).read()[-600:])
PY
}
b() { name=$1; shift; timeout 300 python bench.py "$@" > "$OUT/bench_$name.log" 2>&1; line "$OUT/bench_$name.log"; }
stamp bench
b cfg2_default
b cfg2_group --kernel group --no-cpu-baseline
trace() { name=$1; shift; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$name" -o t -- python "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/trace_$name.log" 2>&1)
  f=$(find "$OUT/trace_$name" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${name}_kernel_stats.csv" && echo "-- $name" && cut -d, -f1-4 "$f" | head -6; rm -rf "$OUT/trace_$name"; }
stamp traces
trace cfg2_default
pmc() { name=$1; shift; (cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_$name" -o p -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline "$@" > "$OUT/pmc_$name.log" 2>&1)
  python - "$OUT" "$name" <<'PY'
import csv, sys, collections, glob, json
out = {}
for f in sorted(glob.glob(sys.argv[1] + "/pmc_" + sys.argv[2] + "/*/p_counter_collection.csv") + glob.glob(sys.argv[1] + "/pmc_" + sys.argv[2] + "/p_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "tile_" in r["Kernel_Name"] or "classify" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void mbk::", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            out.setdefault(k, {}).update(grid=r["Grid_Size"], workgroup=r["Workgroup_Size"], vgpr=r["VGPR_Count"], sgpr=r["SGPR_Count"])
    for k, d in agg.items():
        for c, v in d.items():
            out[k][c] = {"mean": sum(v) / len(v), "n": len(v)}
json.dump(out, open(sys.argv[1] + "/cfg2_" + sys.argv[2] + "_pmc_by_kernel.json", "w"), indent=1)
for k, d in out.items():
    print(" ", k, {c: round(v["mean"]) for c, v in d.items() if isinstance(v, dict)})
PY
  rm -rf "$OUT/pmc_$name"; }
stamp pmc
pmc strict
pmc cycle --opt cycle_detect=1
stamp more
trace cfg2_cycle --opt cycle_detect=1
b cfg3 --workload cfg3 --no-cpu-baseline
b cfg5 --workload cfg5 --no-cpu-baseline
b chunk_l1 --workload chunk_l1 --no-cpu-baseline
b cfg1 --workload cfg1 --no-cpu-baseline
b exterior --workload exterior --no-cpu-baseline
b inset --workload inset --no-cpu-baseline
b cfg2_f32 --precision f32 --no-cpu-baseline
b cfg2_scan --kernel scan --no-cpu-baseline
stamp level; timeout 120 python scripts/level_rate.py 16 1024 > "$OUT/level16.log" 2>&1; grep "level\|two" "$OUT/level16.log"
b cfg4_f32 --workload cfg4 --no-cpu-baseline
stamp done; du -sh "$OUT"
