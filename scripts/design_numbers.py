"""Print the measurement table of DESIGN.md section 5 from the bench lines collected under profiles/<round>/.
    python scripts/design_numbers.py [round]"""
import json, os, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
d = os.path.join("profiles", rnd)


def load(name):
    p = os.path.join(d, f"bench_{name}.json")
    return json.load(open(p)) if os.path.exists(p) else None


rows = [("**cfg2 (headline)**", "cfg2_default"), ("cfg3 8192² deep zoom mrd 10000", "cfg3"), ("cfg5 4096² mrd 5000, ν + count", "cfg5"),
        ("DataChunk (1,0,0) mrd 1000", "chunk_l1"), ("uniform in-set tile (pure loop rate)", "inset"),
        ("all-exterior DataChunk (4,0,0)", "exterior"), ("the same, counts + bytes", "exterior_both"),
        ("cfg1 512² mrd 256 (plumbing size)", "cfg1"),
        ("cfg2 in fp32 (`--precision f32`)", "cfg2_f32"), ("cfg4 16384² seahorse mrd 50000 fp32", "cfg4_f32")]
print("| workload | strict: G pixel-iter/s | launch ms | flops frac | VALU slot util | cycle test: G/s (reference-equivalent), ms, × | CPU port |")
print("|---|---|---|---|---|---|---|")
for label, name in rows:
    r = load(name)
    if not r:
        continue
    ro, cy, cb = r["roofline"], r.get("cycle_detection"), r.get("cpu_baseline")
    cpu = ""
    if cb:
        cpu = f"{cb['value']:.1f} ({cb['cores']} threads)"
        if "best_effort_avx512" in cb:
            cpu += f" / {cb['best_effort_avx512']['value']:.0f} (AVX-512) / {cb['single_thread']['value']:.2f} (1 thread)"
    print(f"| {label} | {r['value']:,.0f} | {ro['kernel_ms_avg']:.4g} | {ro['frac']:.3f} | {ro['valu_slot_util']:.3f} | "
          + (f"{cy['value']:,.0f}, {cy['ms_per_step']:.4g}, ×{cy['speedup_vs_strict']:.2f}" if cy else "") + f" | {cpu} |")
ks = [(k, load(f"cfg2_{k}")) for k in ("group", "scan", "asm", "simple", "refill")]
print("| cfg2, kernel " + " / ".join(k for k, r in ks if r) + " | " + " / ".join(f"{r['value']:,.0f}" for k, r in ks if r) + " | | | | | |")
for name in ("queue_n1", "queue_n2_oversub", "own_n2_oversub", "bands_n1_cfg3", "bands_n2_oversub"):
    r = load(name)
    if r:
        c = r["config"]
        print(f"{name}: {r['value']:,.0f} G/s, {r['ms_per_step']:.3f} ms/step, n={r['n_gpus']}, per rank {c.get('tiles_per_rank', c.get('bands_per_rank'))}, "
              f"once {c.get('tiles_exactly_once', c.get('bands_exactly_once'))}, launches {c.get('launches_per_rank')}")
r = load("cfg2_default")
if r:
    for k in ("two_streams", "end_to_end", "queue_job"):
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in r.get(k, {}).items() if a != "what"})
    print("roofline", {a: (round(b, 4) if isinstance(b, float) else b) for a, b in r["roofline"].items() if a not in ("traffic_source",)})
    print("traffic_source", r["roofline"]["traffic_source"])
