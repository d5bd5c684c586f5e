# shared helpers of the scripts/gpu_r3*.sh passes (sourced; expects ROOT and OUT)
line() { python - "$1" <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); ro=r["roofline"]; cy=r.get("cycle_detection")
    print(f"  {r['config']['workload'][:9]:9s} {r['config']['kernel']:8s} {r['dtype']} n={r['n_gpus']} {r['config']['shard']:5s} {str(r['config'].get('options')):24s} {r['value']:9.1f} G/s  ms/step {r['ms_per_step']:.4f} launch ms avg {ro['kernel_ms_avg']:.4f} med {ro.get('kernel_ms_median',0):.4f} min {ro['kernel_ms_min']:.4f}  frac {ro['frac']:.3f} slot_util {(ro['valu_slot_util'] or 0):.3f}"
          + (f" | cycle test on: {cy['value']:.1f} G/s-eq {cy['ms_per_step']:.4f} ms x{cy['speedup_vs_strict']:.2f} same={cy['same_pixel_iterations_and_never_count']}" if cy else ""))
    for k in ("two_streams", "end_to_end", "queue_job"):
        if k in r: print("    ", k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in r[k].items() if a != "what"})
    c=r["config"]
    if c["shard"] != "own": print("     once", c.get("tiles_exactly_once", c.get("bands_exactly_once")), "per rank", c.get("tiles_per_rank", c.get("bands_per_rank")), "finish ms", c["rank_finish_ms"], "gpus", c["distinct_gpus"], [x["pci_bus_id"] for x in c["ranks_seen"]])
except Exception as e:
    print("  FAILED", sys.argv[1], e); print(open(sys.argv[1]).read()[-800:])
PY
}
b() { name=$1; shift; timeout 900 python bench.py "$@" > "$OUT/bench_$name.log" 2>&1; line "$OUT/bench_$name.log"; }
trace() { name=$1; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$name" -o t -- python "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/trace_$name.log" 2>&1)
  f=$(find "$OUT/trace_$name" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${name}_kernel_stats.csv" && echo "-- $name" && cut -d, -f1-6 "$f" | head -6; rm -rf "$OUT/trace_$name"; }
# one PMC pass (no tracing) of bench.py: pmcrun NAME "COUNTERS" bench-args...
pmcrun() { name=$1; counters=$2; shift 2; (cd /tmp && timeout 600 rocprofv3 --pmc $counters --output-format csv -d "$OUT/pmc_$name" -o p -- python "$ROOT/bench.py" --no-cpu-baseline --no-extras "$@" > "$OUT/pmc_$name.log" 2>&1); line "$OUT/pmc_$name.log"; }
# the legs of a default line that the driver's record is judged on (round 6)
legs() { python - "$1" <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); ro=r["roofline"]
    cy=r.get("cycle_detection") or {}; tw=r.get("two_streams") or {}; su=r.get("sustained") or {}; xb=r.get("xcd_balance_opt_in") or {}; ee=r.get("end_to_end") or {}
    print(f"     headline {r['value']:.1f} G/s {r['ms_per_step']:.4f} ms | per-launch pass avg {(ro.get('kernel_ms_avg_per_launch_pass') or 0):.4f} med {ro.get('kernel_ms_median',0):.4f} min {ro['kernel_ms_min']:.4f} (region avg {ro['kernel_ms_avg']:.4f}) | submit {r['config'].get('host_submit_us_per_launch')} us")
    print(f"     cycle leg {cy.get('ms_per_step')} ms (events {cy.get('kernel_ms_avg')}, steps_run {cy.get('steps_run')}, submit {cy.get('host_submit_us_per_launch')} us) | two_streams {tw.get('ms_per_step')} ms ratio {tw.get('ratio_to_headline_value')}")
    if su: print(f"     sustained {su.get('value')} G/s {su.get('ms_per_step')} ms x{su.get('steps_run')} frac {su.get('roofline_frac')} start {su.get('start')} middle {su.get('middle')} end {su.get('end')}")
    if xb: print(f"     xcd_balance_opt_in {xb.get('value')} ratio {xb.get('ratio_to_headline_value')}")
    if ee: print("     end_to_end", {k: round(v, 1) for k, v in ee.items() if k.startswith("tiles_per_s")})
    for k, v in (r.get("configs") or {}).items(): print(f"     {k}: {v.get('value')} G/s {v.get('ms_per_step')} ms x{v.get('steps_run', v.get('steps'))} frac {(v.get('roofline') or {}).get('frac')} verified {(v.get('output_verified') or {}).get('verified')} {v.get('error','')}")
    print("     verified", (r.get("output_verified") or {}).get("verified"), "versions", r["config"].get("versions"), "smi", r["config"].get("smi_around_timed_region"))
except Exception as e:
    print("  legs FAILED", sys.argv[1], e)
PY
}
