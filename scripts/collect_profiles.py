"""Copy the judged summaries of a gpurun_out/<tag> directory into profiles/<round>/ (tracked)."""
import collections, csv, json, os, shutil, sys

tag, rnd = sys.argv[1], sys.argv[2]
src, dst = os.path.join("gpurun_out", tag), os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "prof_trace", "bench_kernel_stats.csv"), os.path.join(dst, "cfg2_default_kernel_stats.csv"))
for f in sorted(os.listdir(src)):
    if f.startswith("bench") and f.endswith(".log"):
        lines = [l for l in open(os.path.join(src, f)) if l.startswith("{")]
        if lines:
            name = "bench_cfg2_default.json" if f == "bench.log" else f.replace(".log", ".json").replace("bench_", "bench_cfg2_" if f[6:-4] in ("asm", "simple", "refill", "group") else "bench_")
            name = name.replace("bench_cfg2_f32", "bench_cfg2_f32")
            open(os.path.join(dst, name), "w").write(lines[-1])
out = {}
for d in ["prof_pmc", "prof_pmc_w", "prof_pmc_f"]:
    p = os.path.join(src, d, "bench_counter_collection.csv")
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if "tile_" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            out.update(kernel=r["Kernel_Name"], grid=r["Grid_Size"], workgroup=r["Workgroup_Size"],
                       vgpr=r["VGPR_Count"], sgpr=r["SGPR_Count"])
    for k, v in agg.items():
        out[k] = {"mean": sum(v) / len(v), "n": len(v)}
json.dump(out, open(os.path.join(dst, "cfg2_default_pmc_summary.json"), "w"), indent=1)
for extra in ("rocminfo.txt", "nproc.txt", "valu_rates.log"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(dst, extra))
print(json.dumps(out, indent=1))
print(sorted(os.listdir(dst)))
