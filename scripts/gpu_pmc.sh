#!/bin/bash
# PMC counters for one kernel/workload.  Usage: scripts/gpu_pmc.sh tag kernel workload
set -u
TAG=${1:-pmc}; K=${2:-default}; W=${3:-cfg2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python "$ROOT/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --kernel $K --workload $W > "$OUT/$name.log" 2>&1; }
run a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
python3 - "$OUT" <<'PY'
import csv, sys, collections, glob
for f in sorted(glob.glob(sys.argv[1] + "/*/p_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "tile_" not in k: continue
        print(k)
        for c, v in sorted(d.items()):
            print(f"   {c:24s} {sum(v)/len(v):16.0f}")
PY
