"""Copy the judged summaries of a gpurun_out/<tag> directory (written by scripts/gpu_run.sh) into
profiles/<round>/ (tracked).   python scripts/collect_profiles.py <tag> <round>"""
import glob, json, os, shutil, sys

tag, rnd = sys.argv[1], sys.argv[2]
src, dst = os.path.join("gpurun_out", tag), os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)
for f in sorted(glob.glob(os.path.join(src, "bench_*.log"))):
    lines = [l for l in open(f) if l.startswith("{")]
    if lines:
        open(os.path.join(dst, os.path.basename(f).replace(".log", ".json")), "w").write(lines[-1])
for f in sorted(glob.glob(os.path.join(src, "*_kernel_stats.csv"))):
    shutil.copy(f, dst)
for f in sorted(glob.glob(os.path.join(src, "power_*.json"))):
    shutil.copy(f, dst)
for extra in ("rocminfo.txt", "nproc.txt", "valu_rates.log", "level16.log", "worker_e2e.log", "cfg2_default_pmc_by_kernel.json", "soak.log",
              "pytest_gpu.log", "cfg3_group_pmc.json", "cfg3_refill_pmc.json", "source_sha256.txt", "smoke.log",
              "light_path_microbench.txt", "mfma_f64_coissue.txt", "exterior_default_pmc_by_kernel.json", "pytest_gpu.txt", "soak.txt",
              "valu_issue.txt", "scale_prediction.json", "scale_emulate.txt", "cfg3_default_pmc_by_kernel.json",
              "chunk_l1_default_pmc_by_kernel.json"):
    p = os.path.join(src, extra)
    if os.path.exists(p):
        out = extra.replace("valu_rates.log", "valu_rates_microbench.txt").replace("soak.log", "soak.txt").replace("pytest_gpu.log", "pytest_gpu.txt")
        text = "".join(l for l in open(p, errors="replace") if "amdgpu.ids" not in l)
        open(os.path.join(dst, out), "w").write(text)
# the per-launch HBM traffic bench.py quotes: WRITE_SIZE / FETCH_SIZE of the dominant cfg2 kernel
by = os.path.join(src, "cfg2_default_pmc_by_kernel.json")
if os.path.exists(by):
    d = json.load(open(by))
    # the strict kernel (cycle test off) is the one the headline is measured on: of the tile kernels of the run (both legs
    # are in it) the one with the most vector instructions per launch -- the cycle test only ever removes instructions
    dom = max((k for k in d if k.startswith("tile_")), key=lambda k: d[k].get("SQ_INSTS_VALU", {}).get("mean", 0), default=None)
    if dom:
        out = {"kernel": dom}
        # the sources the counters were collected on: bench.py quotes `traffic` only while they match the tree
        sha = os.path.join(src, "source_sha256.txt")
        if os.path.exists(sha):
            out["source_sha256"] = open(sha).read().split()[-1]
        out["note"] = ("vgpr_as_reported is rocprofv3's VGPR_Count = (granulated count + 1) x 4, i.e. half of the registers "
                       "allocated (gfx950 allocates in granules of 8)")
        out.update(d[dom])
        json.dump(out, open(os.path.join(dst, "cfg2_default_pmc_summary.json"), "w"), indent=1)
print(sorted(os.listdir(dst)))
