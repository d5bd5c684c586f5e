#!/bin/bash
# The one parameterised GPU pass (round 4; replaces the one-off gpu_r3?.sh scripts).  Usage, on the GPU box:
#     scripts/gpu_run.sh TAG section [section ...]
# Everything lands in gpurun_out/TAG/.  Sections (run in the order given):
#   smoke        __graft_entry__.smoke()
#   tests        the whole `pytest -m gpu` suite            tests:EXPR  only tests matching -k EXPR
#   soak[:S[:SEED]]  randomized parity soak for S seconds (default 120; seed 17)
#   issue        profiles/microbench/valu_issue.hip (fp64 issue rate in shader cycles)
#   clock[:W:K ...]  the shader clock (s_memtime / s_memrealtime, a one-wave probe in a second process) while bench.py runs workload W for K steps
#   headline     bench.py as the driver runs it
#   balance      MBK_OPT_XCD_BALANCE on / off on the same box (queue job, cfg2, DataChunk (1,0,0))
#   n2           the N > 1 paths with two ranks on this box's one GPU (--oversubscribe; functional)
#   emulate      scripts/scale_emulate.py -> scale_prediction.json
#   benches      the other bench lines (kernels, workloads)
#   ab:NAME=V    same-box A/B of a library option: cfg2 / chunk_l1 / cfg3 with and without --opt NAME=V, twice, interleaved
#   abprev       same-box A/B against an earlier commit's tree built under .ab/prev
#   skew         profiles/microbench/units_skew.hip (uneven H shares per XCD: even / weighted launches alternate)
#   traces       rocprofv3 --kernel-trace --stats of the named workloads
#   pmc          rocprofv3 --pmc passes (cfg2 with the source hash, cfg3)
#   power        power / clock traces
#   e2e          level rate, worker end to end
#   (round 5) strip[:reps] (scripts/strip_ab.py) | fill | gaps:W:OPT+OPT | traceopt:W:OPT+OPT[:args] | extprobe | unitstrace:W,cycle,m_late,h_settled ...
#   (round 6) driverline:N[:args] (the driver's exact command, N times, + wall time) | prevline:N (the same on the tree under .ab/prev) | burst:C ... (burst_probe.py) |
#             anyorder | execrate | classes:W,cycle ... (units_classes.hip + PMC + class_table.py) | spillshard (SPILL's gate under the sharded modes)
set -u
TAG=${1:?tag}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12 > "$OUT/rocminfo.txt" 2>&1; nproc > "$OUT/nproc.txt"
python -c "
import sys; sys.path.insert(0, '.'); import bench; print(bench.kernel_source_hash())" > "$OUT/source_sha256.txt"; cat "$OUT/source_sha256.txt"
for SEC in "$@"; do
  ARG=""; case "$SEC" in *:*) ARG=${SEC#*:}; SEC=${SEC%%:*};; esac
  echo "== $SEC $ARG"
  case "$SEC" in
  smoke) timeout 600 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log";;
  tests) if [ -n "$ARG" ]; then timeout 2400 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider -k "$ARG" > "$OUT/pytest_gpu_subset.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu_subset.txt"; grep -E "^FAILED|^ERROR" "$OUT/pytest_gpu_subset.txt" | cut -c1-300 | head -10
         else timeout 2400 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > "$OUT/pytest_gpu.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.txt"; grep -E "^FAILED|^ERROR" "$OUT/pytest_gpu.txt" | cut -c1-300 | head -10; fi;;
  soak) S=${ARG%%:*}; SEED=${ARG#*:}; [ "$SEED" = "$ARG" ] && SEED=17   # soak[:seconds[:seed]]
        timeout 900 python scripts/gpu_soak.py ${S:-120} $SEED > "$OUT/soak.txt" 2>&1; tail -2 "$OUT/soak.txt";;
  issue) hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue profiles/microbench/valu_issue.hip 2> "$OUT/build_issue.log" && timeout 300 /tmp/valu_issue > "$OUT/valu_issue.txt" 2>&1; cut -c1-200 "$OUT/valu_issue.txt";;
  clock) hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue profiles/microbench/valu_issue.hip 2> "$OUT/build_issue.log"
      for spec in ${ARG:-cfg2:4000 inset:700 cfg3:160 chunk_l1:6000}; do W=${spec%%:*}; K=${spec#*:}
        /tmp/valu_issue --probe 9000 > "$OUT/clock_probe_$W.txt" 2>&1 &
        PROBE=$!
        sleep 1.5
        timeout 300 python bench.py --workload $W --no-cpu-baseline --no-extras --steps $K --opt cycle_detect=0 > "$OUT/bench_clock_$W.log" 2>&1
        wait $PROBE
        line "$OUT/bench_clock_$W.log"
        python - "$OUT/clock_probe_$W.txt" <<'PY'
import sys
rows=[l.split() for l in open(sys.argv[1]) if l[0] not in "#d"]
rows=[(float(a),float(b),float(c)) for a,b,c in rows if len((a,b,c))==3]
print("     t(ms):MHz:probe ms  " + "  ".join(f"{t:.0f}:{m:.0f}:{d:.2f}" for t,m,d in rows[::12]))
PY
      done;;
  headline) b cfg2_default;;
  burst) for C in ${ARG:-0}; do timeout 300 python scripts/burst_probe.py $C 2>&1 | grep -v amdgpu.ids | tee "$OUT/burst_probe_$C.txt"; done;;
  classes) # per-class instruction table of a units launch: classes:W,cycle[,m_late,h_settled] ...
      hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I. -o /tmp/units_classes profiles/microbench/units_classes.hip 2> "$OUT/build_units_classes.log"
      C1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
      C2="SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
      for A in ${ARG:-cfg2,0 cfg2,1}; do N=units_classes_${A//,/_}
        timeout 300 /tmp/units_classes ${A//,/ } > "$OUT/$N.txt" 2>&1
        (cd /tmp && timeout 900 rocprofv3 --pmc $C1 --output-format csv -d "$OUT/pmc_${N}_a" -o p -- /tmp/units_classes ${A//,/ } > "$OUT/pmc_${N}_a.log" 2>&1)
        (cd /tmp && timeout 900 rocprofv3 --pmc $C2 --output-format csv -d "$OUT/pmc_${N}_b" -o p -- /tmp/units_classes ${A//,/ } > "$OUT/pmc_${N}_b.log" 2>&1)
        python scripts/pmc_summary.py "$OUT/${N}_pmc_by_kernel.json" "$OUT/pmc_${N}_a" "$OUT/pmc_${N}_b" --match class_units_kernel > /dev/null
        python scripts/class_table.py "$OUT/$N.txt" "$OUT/${N}_pmc_by_kernel.json" | tee "$OUT/class_table_${A//,/_}.txt"
        rm -rf "$OUT"/pmc_${N}_?
      done;;
  spillshard) # SPILL under the sharded modes (several launches in flight hide the second pass' tail): cfg3 row bands / 2x2 / 4x4 tiles with the gate at 19 (default), 16, 0
      for rep in 1 2; do for G in 19 17 16 0; do
        b cfg3_bands_gate${G}_$rep --workload cfg3 --shard bands --no-cpu-baseline --opt spill_min_blocks=$G
        b cfg3_grid2_gate${G}_$rep --workload cfg3 --shard queue --grid 2 --steps 6 --no-cpu-baseline --opt spill_min_blocks=$G
        b cfg3_grid4_gate${G}_$rep --workload cfg3 --shard queue --grid 4 --steps 3 --no-cpu-baseline --opt spill_min_blocks=$G
      done; done;;
  execrate) hipcc --offload-arch=gfx950 -O3 -o /tmp/exec_rate profiles/microbench/exec_rate.hip 2> "$OUT/build_exec_rate.log" && timeout 120 /tmp/exec_rate > "$OUT/exec_rate.txt" 2>&1; cat "$OUT/exec_rate.txt";;
  anyorder) hipcc --offload-arch=gfx950 -O3 -o /tmp/anyorder profiles/microbench/anyorder.hip 2> "$OUT/build_anyorder.log" && timeout 120 /tmp/anyorder > "$OUT/anyorder.txt" 2>&1; cat "$OUT/anyorder.txt";;
  driverline) # the driver's exact command (round 6), ARG times; driverline:N[:extra bench args with commas for spaces]
      N=${ARG%%:*}; X=${ARG#*:}; [ "$X" = "$ARG" ] && X=""; for rep in $(seq 1 ${N:-1}); do
        T0=$(date +%s.%N); python bench.py --gpus 1 --steps 20 --warmup 5 ${X//,/ } > "$OUT/bench_driverline_$rep.log" 2>&1; T1=$(date +%s.%N)
        line "$OUT/bench_driverline_$rep.log"; legs "$OUT/bench_driverline_$rep.log"; echo "     wall $(python -c "print(round($T1 - $T0, 2))") s" | tee "$OUT/driverline_$rep.time"; done;;
  prevline) # the same command on the tree under .ab/prev (an earlier commit, built there)
      for rep in $(seq 1 ${ARG:-1}); do
        python .ab/prev/bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_prevline_$rep.log" 2>&1
        line "$OUT/bench_prevline_$rep.log"; legs "$OUT/bench_prevline_$rep.log"; done;;
  balance) # MBK_OPT_XCD_BALANCE on (default) / off, same box: the queue job (4 tiles in flight) and the own-mode legs of cfg2 and DataChunk (1,0,0)
      for rep in 1 2; do
        b queue_n1_balance_$rep --shard queue --no-cpu-baseline
        b queue_n1_even_$rep --shard queue --no-cpu-baseline --opt xcd_balance=0
        for W in cfg2 chunk_l1; do
          b ${W}_balance_$rep --workload $W --no-cpu-baseline --no-extras
          b ${W}_even_$rep --workload $W --no-cpu-baseline --no-extras --opt xcd_balance=0
        done; done;;
  n2) b queue_n1 --shard queue --no-cpu-baseline
      b queue_n2_oversub --gpus 2 --oversubscribe
      b bands_n2_oversub --gpus 2 --oversubscribe --shard bands --workload cfg3 --steps 6
      b cfg3_queue_n2_oversub --gpus 2 --oversubscribe --workload cfg3 --grid 2 --steps 6
      for f in queue_n2_oversub bands_n2_oversub cfg3_queue_n2_oversub; do python - "$OUT/bench_$f.log" <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); s=r.get("single_gpu_same_job",{})
    print("     same job on one GPU:", round(s.get("value",0),1), "G/s  speed-up", round(r.get("speedup_same_job",0),3), "efficiency", round(r.get("efficiency_same_job",0),3), s.get("error",""))
except Exception as e: print("     FAILED", e)
PY
      done;;
  emulate) timeout 1500 python scripts/scale_emulate.py --out "$OUT/scale_prediction.json" ${ARG:+--jobs $ARG} > "$OUT/scale_emulate.txt" 2>&1; grep -v amdgpu.ids "$OUT/scale_emulate.txt" | tail -24;;
  benches)
      for K in group scan asm simple refill; do b cfg2_$K --kernel $K --no-cpu-baseline --no-extras; done
      b cfg2_cycle --opt cycle_detect=1 --no-cpu-baseline --no-extras
      b cfg1 --workload cfg1 --no-cpu-baseline
      b exterior --workload exterior --no-cpu-baseline; b exterior_both --workload exterior --outputs both --no-cpu-baseline
      b chunk_l1 --workload chunk_l1 --no-cpu-baseline; b inset --workload inset --no-cpu-baseline
      b cfg3 --workload cfg3; b cfg5 --workload cfg5; b cfg2_f32 --precision f32 --no-cpu-baseline
      b cfg4_f32 --workload cfg4;;
  wavelimit) # same-box sweep of MBK_OPT_WAVE_LIMIT (resident waves per SIMD of the one-wave-per-block kernels)
      for W in exterior cfg2 chunk_l1 inset cfg3; do for L in 0 6 4 3 2 0; do
        K=group; [ $W = cfg3 ] && X="--steps 12 --warmup 2" || X=""
        b wl_${W}_$L --workload $W --kernel $K --no-cpu-baseline --no-extras --opt wave_limit=$L $X
      done; done;;
  pmckernel) # PMC + trace of one kernel selector on one workload: pmckernel:KERNEL:WORKLOAD
      K=${ARG%%:*}; W=${ARG#*:}
      C1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
      C2="SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
      trace ${W}_${K} --workload $W --kernel $K --no-extras
      pmcrun ${W}_${K}_a "$C1" --workload $W --kernel $K --steps 20 --warmup 5; pmcrun ${W}_${K}_b "$C2" --workload $W --kernel $K --steps 20 --warmup 5
      python scripts/pmc_summary.py "$OUT/${W}_${K}_pmc_by_kernel.json" "$OUT/pmc_${W}_${K}_a" "$OUT/pmc_${W}_${K}_b" --match tile_
      rm -rf "$OUT"/pmc_*_?;;
  pmcopt) # PMC (instruction counts, busy, occupancy) of the default kernel with and without an option: pmcopt:NAME=V[:workloads]
      O=${ARG%%:*}; WL=${ARG#*:}; [ "$WL" = "$ARG" ] && WL="cfg2 chunk_l1 cfg3"
      C1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
      C2="SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
      for W in $WL; do case $W in cfg3) X="--steps 4 --warmup 1";; *) X="--steps 20 --warmup 5";; esac
        for V in base opt; do [ $V = opt ] && OO="--opt $O" || OO=""
          pmcrun ${W}_${V}_a "$C1" --workload $W $X $OO --opt cycle_detect=0; pmcrun ${W}_${V}_b "$C2" --workload $W $X $OO --opt cycle_detect=0
          python scripts/pmc_summary.py "$OUT/${W}_${V}_pmc_by_kernel.json" "$OUT/pmc_${W}_${V}_a" "$OUT/pmc_${W}_${V}_b" --match tile_ | cut -c1-600
        done; done
      rm -rf "$OUT"/pmc_*_?;;
  unitstrace) hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I. -o /tmp/units_trace profiles/microbench/units_trace.hip 2> "$OUT/build_units_trace.log"
      for A in ${ARG:-cfg2,0,0,0 chunk_l1,0,0,0}; do   # workload,cycle test,m_late,h_settled[,cycle window (default 32)]
        set -- ${A//,/ }; W=$1; N=units_trace_${A//,/_}
        timeout 300 /tmp/units_trace $W "$OUT/$N.bin" ${2:-0} ${3:-0} ${4:-0} ${5:-32} > "$OUT/$N.txt" 2>&1
        timeout 600 python scripts/analyze_units_trace.py "$OUT/$N.bin" >> "$OUT/$N.txt" 2>&1
        grep -v "^  *[0-9.]* *[0-9.]* *[0-9.]* *[0-9.]* *[0-9.]*$" "$OUT/$N.txt" | tail -48; rm -f "$OUT/$N.bin"
      done;;
  extprobe) # the host-buffer pipeline on all-exterior tiles: rate by slots, then the HIP calls and the GPU side of it
      for K in 1 2 4; do timeout 120 python scripts/exterior_pipeline_probe.py 2000 $K 1 2>&1 | grep -v amdgpu.ids; done | tee "$OUT/exterior_pipeline.txt"
      timeout 120 python scripts/exterior_pipeline_probe.py 2000 4 0 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/exterior_pipeline.txt"
      (cd /tmp && timeout 600 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d "$OUT/extprobe" -o t -- python "$ROOT/scripts/exterior_pipeline_probe.py" 2000 4 1 > "$OUT/extprobe.log" 2>&1)
      tail -1 "$OUT/extprobe.log" | tee -a "$OUT/exterior_pipeline.txt"
      for f in hip_api_stats kernel_stats memory_copy_stats; do g=$(find "$OUT/extprobe" -name "*${f}.csv" | head -1); [ -n "$g" ] && { echo "-- $f"; cut -d, -f1-6 "$g" | head -14; cp "$g" "$OUT/extprobe_${f}.csv"; }; done | tee -a "$OUT/exterior_pipeline.txt"
      python - "$OUT/extprobe" <<'PY' | tee -a "$OUT/exterior_pipeline.txt"
import csv, glob, os, sys
import numpy as np
p = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
if p:
    rows = list(csv.DictReader(open(p[0])))
    nm = lambda r: r.get("Kernel_Name") or r.get("Name")
    light = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "tile_light_kernel" in nm(r))
    if len(light) > 100:
        s = np.array([a for a, b in light][len(light) // 4:], float); e = np.array([b for a, b in light][len(light) // 4:], float)
        print(f"light kernels: {len(light)}; duration mean {np.mean(e - s) / 1e3:.1f} us; start-to-start mean {np.mean(np.diff(s)) / 1e3:.1f} us (median {np.median(np.diff(s)) / 1e3:.1f}); overlapping the previous one: {np.mean(s[1:] < e[:-1]):.2f}")
PY
      rm -rf "$OUT/extprobe";;
  traceopt) # kernel-trace stats of one workload under options: traceopt:W:OPT+OPT[:extra bench args with commas for spaces]
      W=${ARG%%:*}; R=${ARG#*:}; O=${R%%:*}; X=${R#*:}; [ "$X" = "$R" ] && X=""; OPTS=""; for o in ${O//+/ }; do OPTS="$OPTS --opt $o"; done
      trace ${W}_${O//[=+]/} --workload $W --no-extras $OPTS ${X//,/ }; f="$OUT/${W}_${O//[=+]/}_kernel_stats.csv"; [ -f "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if r[0] != "Name": print(f"     {r[0][:70]:70s} calls {r[1]:>6s} avg {float(r[3]) / 1e3:10.1f} us  total {float(r[2]) / 1e6:9.1f} ms")
PY
      ;;
  gaps) # kernel-trace of back-to-back steps: tile kernel, gap, where the next launch's pre-pass ran.  gaps:W:OPT+OPT...
      W=${ARG%%:*}; O=${ARG#*:}; [ "$O" = "$ARG" ] && O=""; OPTS=""; for o in ${O//+/ }; do OPTS="$OPTS --opt $o"; done
      N=gaps_${W}_${O//[=+]/}
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/$N" -o t -- python "$ROOT/bench.py" --no-cpu-baseline --no-extras --workload $W --steps 150 --warmup 20 $OPTS > "$OUT/$N.log" 2>&1)
      line "$OUT/$N.log"; python scripts/analyze_gaps.py "$OUT/$N" | tee "$OUT/$N.txt"; rm -rf "$OUT/$N";;
  strip) timeout 400 python scripts/strip_ab.py ${ARG:-5} 2>&1 | grep -v amdgpu.ids | tee "$OUT/strip_ab.txt";;
  fill) hipcc --offload-arch=gfx950 -O3 -o /tmp/fill profiles/microbench/fill.hip 2> "$OUT/build_fill.log" && timeout 300 /tmp/fill > "$OUT/fill.txt" 2>&1; cat "$OUT/fill.txt"
      b exterior_fillbox --workload exterior --no-cpu-baseline --no-extras; b exterior_both_fillbox --workload exterior --outputs both --no-cpu-baseline --no-extras;;
  skew) [ -x build/units_skew ] || hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I. -o build/units_skew profiles/microbench/units_skew.hip 2> "$OUT/build_units_skew.log"
      for A in ${ARG:-cfg2,40,0.5 chunk_l1,40,0.5 cfg2,40,0.5,0,1,0 cfg2,40,0.5,0,1,1}; do   # workload,launches,gain[,rotation[,cycle test[,signal]]]
        timeout 120 build/units_skew ${A//,/ } > "$OUT/units_skew_${A//,/_}.txt" 2>&1; tail -4 "$OUT/units_skew_${A//,/_}.txt"
      done;;
  events) for rep in 1 2; do for W in cfg2 chunk_l1; do
        b ${W}_region_$rep --workload $W --no-cpu-baseline --no-extras --launch-events region
        b ${W}_perlaunch_$rep --workload $W --no-cpu-baseline --no-extras --launch-events per-launch
      done; done;;
  wg4) for rep in 1 2; do
        b cfg3_wg1_$rep --workload cfg3 --kernel group --no-cpu-baseline --no-extras --steps 12 --warmup 2
        b cfg3_wg4_$rep --workload cfg3 --kernel group --no-cpu-baseline --no-extras --steps 12 --warmup 2 --opt waves_per_wg=4
        b cfg2_wg1_$rep --kernel group --no-cpu-baseline --no-extras
        b cfg2_wg4_$rep --kernel group --no-cpu-baseline --no-extras --opt waves_per_wg=4
      done;;
  ab) for rep in 1 2; do for W in ${ABW:-cfg2 chunk_l1 cfg3}; do
        b ${W}_base_$rep --workload $W --no-cpu-baseline --no-extras
        OPTS=""; for o in ${ARG//+/ }; do OPTS="$OPTS --opt $o"; done     # ab:a=1+b=2 sets both
        b ${W}_${ARG//[=+]/}_$rep --workload $W --no-cpu-baseline --no-extras $OPTS
      done; done;;
  abprev) # same-box A/B of the tree against a copy of an earlier commit built under .ab/prev (git archive REV bench.py distributedmandelbrot_amd include oracle)
      for rep in 1 2; do for W in ${ABW:-cfg2 chunk_l1}; do
        timeout 900 python .ab/prev/bench.py --workload $W --no-cpu-baseline --no-extras > "$OUT/bench_${W}_prev_$rep.log" 2>&1; line "$OUT/bench_${W}_prev_$rep.log"
        b ${W}_now_$rep --workload $W --no-cpu-baseline --no-extras
        python - "$OUT/bench_${W}_now_$rep.log" <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print("     xcd_shares", r["config"].get("xcd_shares"))
PY
      done; done;;
  traces) for W in ${ARG:-cfg2 chunk_l1 cfg3 exterior}; do
        case $W in cfg3) X="--steps 10 --warmup 2";; cfg4) X="--steps 2 --warmup 1";; *) X="";; esac
        trace ${W}_default --workload $W --no-extras $X; done;;
  pmc) C1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
       C2="SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
       pmcrun cfg2_a "$C1" --steps 20 --warmup 5; pmcrun cfg2_b "$C2" --steps 20 --warmup 5
       pmcrun cfg2_w "WRITE_SIZE" --steps 20 --warmup 5; pmcrun cfg2_f "FETCH_SIZE" --steps 20 --warmup 5
       python scripts/pmc_summary.py "$OUT/cfg2_default_pmc_by_kernel.json" "$OUT/pmc_cfg2_a" "$OUT/pmc_cfg2_b" "$OUT/pmc_cfg2_w" "$OUT/pmc_cfg2_f" --match tile_
       [ "$ARG" = cfg2 ] && ARG=" "   # pmc:cfg2 = the headline workload only
       for W in ${ARG:-cfg3 chunk_l1}; do
         case $W in cfg3) X="--steps 4 --warmup 1";; *) X="--steps 20 --warmup 5";; esac
         pmcrun ${W}_a "$C1" --workload $W $X; pmcrun ${W}_b "$C2" --workload $W $X
         python scripts/pmc_summary.py "$OUT/${W}_default_pmc_by_kernel.json" "$OUT/pmc_${W}_a" "$OUT/pmc_${W}_b" --match tile_
       done
       rm -rf "$OUT"/pmc_*_?;;
  power) for spec in "cfg3 default 150" "inset default 600" "cfg2 default 4000"; do set -- $spec
        timeout 300 python scripts/power_trace.py "$OUT/power_$1_$2.json" -- python bench.py --workload $1 --kernel $2 --no-cpu-baseline --no-extras --steps $3 --opt cycle_detect=0 > "$OUT/power_$1_$2.log" 2>&1
        python - "$OUT/power_$1_$2.json" <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); print("  ", r["bench"]["workload"][:8], r["bench"]["kernel"], "cap", r.get("power_cap_W"), "busy W p50", r.get("busy_power_W",{}).get("p50"), "sclk p50", r.get("busy_sclk_MHz",{}).get("p50"), "J/Gpi", round(r.get("J_per_G_pixel_iteration",0),4), "G/s", round(r["bench"]["value"],1))
except Exception as e: print("  power FAILED", e)
PY
      done;;
  e2e) timeout 200 python scripts/level_rate.py 16 1024 > "$OUT/level16.log" 2>&1; grep -v amdgpu.ids "$OUT/level16.log"
       timeout 900 python scripts/worker_e2e.py 12 256 3 > "$OUT/worker_e2e.log" 2>&1; grep -v amdgpu.ids "$OUT/worker_e2e.log" | tail -14;;
  *) echo "unknown section $SEC";;
  esac
done
du -sh "$OUT"
