"""Per-class table of a units launch: profiles/microbench/units_classes.hip's own table (entries, exact work
per class) joined with the per-kernel means of its `rocprofv3 --pmc` passes (scripts/pmc_summary.py --match class_units_kernel).
    python scripts/class_table.py units_classes.txt pmc_by_kernel.json
ideal = pixel-iterations / 64 x slots (6.125 per step in 16-step groups: H, settled H; 6.25 in 8-step groups: late M, M; V units:
the light path, no per-step figure -- its instructions are all "overhead" here); lock-step = the same with the wave-steps a wave
runs until its longest pixel is done (strict loops; with the cycle test the executed steps are fewer than either figure)."""
import json
import re
import sys

txt, pmc = open(sys.argv[1]).read().splitlines(), json.load(open(sys.argv[2]))
names = ["all", "late M", "H", "settled H", "M", "V units"]
slots = {"late M": 6.25, "H": 6.125, "settled H": 6.125, "M": 6.25, "V units": 0.0}
rows = {}
for line in txt:
    for nm in names:
        if line.startswith(nm + " ") or line.startswith(nm.ljust(10)):
            f = line[len(nm):].split()
            if len(f) == 5:
                rows[nm] = dict(entries=int(f[0]), iters=int(f[1]), wsteps=int(f[2]), activity=float(f[3]), ms=float(f[4]))
cyc = any("cycle test 1" in l for l in txt[:3])
by_class = {}
for k, v in pmc.items():
    m = re.search(r"class_units_kernel<(true|false), (\d)>", k)
    if m:
        by_class[names[int(m.group(2))]] = v
print(txt[0])
print(f"{'class':10s} {'entries':>8s} {'G px-iter':>10s} {'ideal Minstr':>12s} {'lock-step':>10s} {'measured':>10s} {'excess':>8s} {'= diverg.':>9s} {'+ overhead':>10s} {'lane act.':>9s} {'VALU-busy':>9s} {'us alone':>9s}")
tot = dict(ideal=0.0, lock=0.0, meas=0.0)
for nm in names[1:] + ["all"]:
    r, p = rows.get(nm), by_class.get(nm)
    if not r or not r["entries"]:
        continue
    if nm == "all":
        ideal, lock = tot["ideal"], tot["lock"]
    else:
        ideal, lock = r["iters"] / 64.0 * slots[nm] / 1e6, r["wsteps"] * slots[nm] / 1e6
        tot["ideal"] += ideal
        tot["lock"] += lock
    meas = p["SQ_INSTS_VALU"]["mean"] / 1e6 if p and "SQ_INSTS_VALU" in p else float("nan")
    la = p.get("lane_activity") if p else None
    vb = p.get("valu_busy") if p else None
    print(f"{nm:10s} {r['entries']:8d} {r['iters'] / 1e9:10.4f} {ideal:12.2f} {lock:10.2f} {meas:10.2f} {meas - ideal:8.2f} {lock - ideal:9.2f} {meas - lock:10.2f} "
          f"{(la if la is not None else float('nan')):9.3f} {(vb if vb is not None else float('nan')):9.3f} {r['ms'] * 1e3:9.1f}")
if cyc:
    print("(cycle test on: the measured instructions are below the strict ideal because settled orbits are retired early; compare the classes with each other)")
