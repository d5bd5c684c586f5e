"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: mean of every counter, dispatch count, and -- when the
CSV carries Start/End timestamps -- the mean dispatch duration, the effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration:
the counter is summed over the eight XCDs of an MI355X),
VALU-busy (SQ_ACTIVE_INST_VALU x 4 / (SIMDs x GRBM_GUI_ACTIVE / 8)) and lane activity (SQ_THREAD_CYCLES_VALU / (64 x
SQ_INSTS_VALU... in quad-cycle units: / (16 x SQ_ACTIVE_INST_VALU)).
    python scripts/pmc_summary.py out.json dir_or_csv [dir_or_csv ...] [--match substr]"""
import collections, csv, glob, json, os, sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = None
if "--match" in sys.argv:
    match = sys.argv[sys.argv.index("--match") + 1]
    args = [a for a in args if a != match]
out_path, srcs = args[0], args[1:]
files = []
for s in srcs:
    files += [s] if s.endswith(".csv") else sorted(glob.glob(os.path.join(s, "**", "*counter_collection.csv"), recursive=True))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in files:
    with open(f) as fh:
        rd = csv.DictReader(fh)
        for r in rd:
            name = r["Kernel_Name"].split("(")[0].replace("void mbk::", "")
            if match and match not in name:
                continue
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta.setdefault(name, {}).update(grid=r.get("Grid_Size"), workgroup=r.get("Workgroup_Size"),
                                             vgpr_as_reported=r.get("VGPR_Count"), sgpr=r.get("SGPR_Count"))
            st, en = r.get("Start_Timestamp"), r.get("End_Timestamp")
            if st and en and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                agg[name]["_duration_ns"].append(float(en) - float(st))
out = {}
for name, d in agg.items():
    e = dict(meta[name])
    for c, v in d.items():
        e[c] = {"mean": sum(v) / len(v), "n": len(v)}
    g = e.get("GRBM_GUI_ACTIVE", {}).get("mean")
    dur = e.get("_duration_ns", {}).get("mean")
    act = e.get("SQ_ACTIVE_INST_VALU", {}).get("mean")
    if g and dur:
        e["effective_clock_GHz"] = g / 8.0 / dur   # GRBM_GUI_ACTIVE is the sum over the 8 XCDs (round 3 printed 18.3 "GHz")
    if g and act:
        e["valu_busy"] = act * 4.0 / (1024.0 * g / 8.0)
    thr, ins = e.get("SQ_THREAD_CYCLES_VALU", {}).get("mean"), e.get("SQ_INSTS_VALU", {}).get("mean")
    if thr and act:
        e["lane_activity"] = thr / (64.0 * act)   # both in quad-cycles of issue
    out[name] = e
json.dump(out, open(out_path, "w"), indent=1)
for name, e in out.items():
    print(" ", name[:70], {k: (round(v["mean"]) if isinstance(v, dict) else (round(v, 4) if isinstance(v, float) else v)) for k, v in e.items()})
