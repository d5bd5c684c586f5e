#!/bin/bash
# The round's full GPU evidence pass (everything profiles/<round>/ is built from).  Usage: scripts/gpu_round.sh TAG
# Sections can be skipped with SKIP="tests micro power e2e pmc" (space-separated).
set -u
TAG=${1:-round}; SKIP=" ${SKIP:-} "
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
skip() { [[ "$SKIP" == *" $1 "* ]]; }
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12 > "$OUT/rocminfo.txt" 2>&1; nproc > "$OUT/nproc.txt"
echo "== smoke"; timeout 600 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
if ! skip tests; then echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=6 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"; grep -E "^FAILED|^ERROR" "$OUT/pytest_gpu.log" | cut -c1-200 | head -8; fi
line() { python - "$1" <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); ro=r["roofline"]; cy=r.get("cycle_detection")
    print(f"  {r['config']['workload'][:9]:9s} {r['config']['kernel']:8s} {r['dtype']} {str(r['config'].get('options')):22s} {r['value']:9.1f} G/s  launch ms avg {ro['kernel_ms_avg']:.4f} min {ro['kernel_ms_min']:.4f}  frac {ro['frac']:.3f} slot_util {(ro['valu_slot_util'] or 0):.3f}"
          + (f" | cycle test on: {cy['value']:.1f} G/s-eq {cy['ms_per_step']:.4f} ms x{cy['speedup_vs_strict']:.2f} same={cy['same_pixel_iterations_and_never_count']}" if cy else ""))
except Exception as e:
    print("  FAILED", sys.argv[1], e); print(open(sys.argv[1]).read()[-600:])
PY
}
b() { name=$1; shift; timeout 600 python bench.py "$@" > "$OUT/bench_$name.log" 2>&1; line "$OUT/bench_$name.log"; }
echo "== bench (headline first, with the CPU baseline)"
b cfg2_default
for K in group scan asm simple refill; do b cfg2_$K --kernel $K --no-cpu-baseline; done
b cfg2_group8 --kernel group --opt group_steps=8 --no-cpu-baseline
b cfg2_cycle --opt cycle_detect=1 --no-cpu-baseline
b cfg1 --workload cfg1 --no-cpu-baseline; b cfg1_group --workload cfg1 --kernel group --no-cpu-baseline
b exterior --workload exterior --no-cpu-baseline; b exterior_group --workload exterior --kernel group --no-cpu-baseline
b exterior_both --workload exterior --outputs both --no-cpu-baseline; b exterior_both_group --workload exterior --outputs both --kernel group --no-cpu-baseline
b chunk_l1 --workload chunk_l1 --no-cpu-baseline; b inset --workload inset --no-cpu-baseline
b cfg3 --workload cfg3; b cfg3_refill --workload cfg3 --kernel refill --no-cpu-baseline; b cfg3_scan --workload cfg3 --kernel scan --no-cpu-baseline
b cfg5 --workload cfg5; b cfg2_f32 --precision f32 --no-cpu-baseline
b cfg4_f32 --workload cfg4
echo "== rocprofv3 kernel traces"
trace() { name=$1; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$name" -o t -- python "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/trace_$name.log" 2>&1)
  f=$(find "$OUT/trace_$name" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${name}_kernel_stats.csv" && echo "-- $name" && cut -d, -f1-6 "$f" | head -5; rm -rf "$OUT/trace_$name"; }
trace cfg2_default
trace cfg2_cycle --opt cycle_detect=1
trace exterior_default --workload exterior
trace exterior_both --workload exterior --outputs both
trace chunk_l1_default --workload chunk_l1
trace cfg1_default --workload cfg1
trace cfg3_default --workload cfg3 --steps 10 --warmup 2
trace cfg5_default --workload cfg5 --steps 20 --warmup 3
trace cfg4_f32 --workload cfg4 --steps 2 --warmup 1
if ! skip pmc; then echo "== rocprofv3 pmc (separate passes, no tracing)"
  # bench.py runs the strict leg and the cycle-test leg in one process: the strict counters are those of
  # tile_asm_kernel<double, true, 16, false>, the cycle-test ones of <..., true>
  pmc() { name=$1; shift; (cd /tmp && timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o p -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/pmc_$name.log" 2>&1); }
  pmc a SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
  pmc b SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH
  pmc w WRITE_SIZE; pmc f FETCH_SIZE
  python - "$OUT" <<'PY'
import csv, sys, collections, glob, json
out = {}
for f in sorted(glob.glob(sys.argv[1] + "/pmc_*/*/p_counter_collection.csv") + glob.glob(sys.argv[1] + "/pmc_*/p_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "tile_" in r["Kernel_Name"] or "classify" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void mbk::", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            out.setdefault(k, {}).update(grid=r["Grid_Size"], workgroup=r["Workgroup_Size"], vgpr=r["VGPR_Count"], sgpr=r["SGPR_Count"])
    for k, d in agg.items():
        for c, v in d.items():
            out[k][c] = {"mean": sum(v) / len(v), "n": len(v)}
json.dump(out, open(sys.argv[1] + "/cfg2_default_pmc_by_kernel.json", "w"), indent=1)
for k, d in out.items():
    print(" ", k, {c: round(v["mean"]) for c, v in d.items() if isinstance(v, dict)})
PY
  rm -rf "$OUT"/pmc_a "$OUT"/pmc_b "$OUT"/pmc_w "$OUT"/pmc_f
fi
if ! skip power; then echo "== power traces"
  for spec in "cfg3 group 150" "cfg3 refill 150" "inset default 600" "cfg2 default 4000" "exterior default 8000" "cfg4 default 8"; do set -- $spec
    timeout 300 python scripts/power_trace.py "$OUT/power_$1_$2.json" -- python bench.py --workload $1 --kernel $2 --no-cpu-baseline --steps $3 --opt cycle_detect=0 > "$OUT/power_$1_$2.log" 2>&1
    python - "$OUT/power_$1_$2.json" <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); print("  ", r["bench"]["workload"][:8], r["bench"]["kernel"], "cap", r.get("power_cap_W"), "busy W p50", r.get("busy_power_W",{}).get("p50"), "sclk p50", r.get("busy_sclk_MHz",{}).get("p50"), "J/Gpi", round(r.get("J_per_G_pixel_iteration",0),4), "G/s", round(r["bench"]["value"],1))
except Exception as e: print("  power FAILED", e)
PY
  done
  # the library default (cycle test on): same tile, energy per reference-equivalent pixel-iteration
  timeout 300 python scripts/power_trace.py "$OUT/power_cfg2_cycle.json" -- python bench.py --workload cfg2 --no-cpu-baseline --steps 8000 --opt cycle_detect=1 > "$OUT/power_cfg2_cycle.log" 2>&1
  python -c "import json,sys; r=json.load(open(sys.argv[1])); print('   cfg2 cycle test on: busy W p50', r.get('busy_power_W',{}).get('p50'), 'sclk p50', r.get('busy_sclk_MHz',{}).get('p50'), 'J/Gpi-eq', round(r.get('J_per_G_pixel_iteration',0),4), 'G/s-eq', round(r['bench']['value'],1))" "$OUT/power_cfg2_cycle.json" || echo "  power cycle FAILED"
fi
if ! skip micro; then echo "== microbench"; hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates profiles/microbench/valu_rates.hip && timeout 400 /tmp/valu_rates > "$OUT/valu_rates.log" 2>&1; grep "waves/SIMD=8" "$OUT/valu_rates.log" | cut -c1-60,150-200; fi
if ! skip soak; then echo "== randomized parity soak"; timeout 200 python scripts/gpu_soak.py ${SOAK_S:-60} 11 > "$OUT/soak.log" 2>&1; tail -2 "$OUT/soak.log"; fi
if ! skip e2e; then echo "== level rate / worker end to end"; timeout 200 python scripts/level_rate.py 16 1024 > "$OUT/level16.log" 2>&1; grep "level\|two" "$OUT/level16.log"
  timeout 400 python scripts/worker_e2e.py 12 256 3 > "$OUT/worker_e2e.log" 2>&1; cat "$OUT/worker_e2e.log" | grep -v amdgpu.ids; fi
du -sh "$OUT"
