#!/bin/bash
# Round 3's full GPU evidence pass (everything profiles/r03/ is built from).  Usage: scripts/gpu_round3.sh TAG
# Sections can be skipped with SKIP="tests soak power e2e pmc n2" (space-separated).
set -u
TAG=${1:-r3f}; SKIP=" ${SKIP:-} "
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
skip() { [[ "$SKIP" == *" $1 "* ]]; }
source scripts/gpu_lib.sh
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12 > "$OUT/rocminfo.txt" 2>&1; nproc > "$OUT/nproc.txt"
python -c "
import sys; sys.path.insert(0, '.'); import bench; print(bench.kernel_source_hash())" > "$OUT/source_sha256.txt"; cat "$OUT/source_sha256.txt"
echo "== smoke"; timeout 600 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
if ! skip tests; then echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"; grep -E "^FAILED|^ERROR" "$OUT/pytest_gpu.log" | cut -c1-220 | head -10; fi
if ! skip soak; then echo "== randomized parity soak"; timeout 300 python scripts/gpu_soak.py ${SOAK_S:-120} 17 > "$OUT/soak.log" 2>&1; tail -2 "$OUT/soak.log"; fi
echo "== bench (headline first: the driver's plain command)"
b cfg2_default
for K in group scan asm simple refill; do b cfg2_$K --kernel $K --no-cpu-baseline --no-extras; done
b cfg2_cycle --opt cycle_detect=1 --no-cpu-baseline --no-extras
b cfg1 --workload cfg1 --no-cpu-baseline
b exterior --workload exterior --no-cpu-baseline; b exterior_both --workload exterior --outputs both --no-cpu-baseline
b exterior_group --workload exterior --kernel group --no-cpu-baseline
b chunk_l1 --workload chunk_l1 --no-cpu-baseline; b inset --workload inset --no-cpu-baseline
b cfg3 --workload cfg3; b cfg3_refill --workload cfg3 --kernel refill --no-cpu-baseline
b cfg5 --workload cfg5; b cfg2_f32 --precision f32 --no-cpu-baseline
b cfg4_f32 --workload cfg4
if ! skip n2; then echo "== N > 1 path: the plain command, two ranks sharing this box's one GPU (functional, --oversubscribe)"
  b queue_n1 --shard queue --no-cpu-baseline
  b queue_n2_oversub --gpus 2 --oversubscribe
  b own_n2_oversub --gpus 2 --oversubscribe --shard own --steps 100
  b bands_n2_oversub --gpus 2 --oversubscribe --shard bands --workload cfg3 --steps 4
  b bands_n1_cfg3 --shard bands --workload cfg3 --steps 6 --no-cpu-baseline
fi
echo "== round 3, second half: light pass (finish-in-place against the two-pass form), 32-step groups, microbenchmarks"
b exterior_twopass --workload exterior --no-cpu-baseline --no-extras --opt scan_inline=0
b cfg2_g32 --no-cpu-baseline --no-extras --opt group_steps=32
b cfg3_g32 --workload cfg3 --no-cpu-baseline --no-extras --opt group_steps=32
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I. -o /tmp/light_path profiles/microbench/light_path.hip 2> "$OUT/build_mb.log" && timeout 300 /tmp/light_path > "$OUT/light_path_microbench.txt" 2>&1; cat "$OUT/light_path_microbench.txt"
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/mfma_f64 profiles/microbench/mfma_f64_coissue.hip 2>> "$OUT/build_mb.log" && timeout 300 /tmp/mfma_f64 > "$OUT/mfma_f64_coissue.txt" 2>&1; grep -E "V0|V1  zi|V5|V6" "$OUT/mfma_f64_coissue.txt" | head -4
echo "== rocprofv3 kernel traces"
trace cfg2_default --no-extras
trace cfg2_cycle --opt cycle_detect=1 --no-extras
trace exterior_default --workload exterior
trace exterior_both --workload exterior --outputs both
trace chunk_l1_default --workload chunk_l1
trace cfg1_default --workload cfg1
trace cfg3_default --workload cfg3 --steps 10 --warmup 2
trace cfg5_default --workload cfg5 --steps 20 --warmup 3
trace cfg4_f32 --workload cfg4 --steps 2 --warmup 1
if ! skip pmc; then echo "== rocprofv3 pmc (separate passes, no tracing)"
  C1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
  C2="SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
  pmcrun cfg2_a "$C1" --steps 20 --warmup 5; pmcrun cfg2_b "$C2" --steps 20 --warmup 5
  pmcrun cfg2_w "WRITE_SIZE" --steps 20 --warmup 5; pmcrun cfg2_f "FETCH_SIZE" --steps 20 --warmup 5
  python scripts/pmc_summary.py "$OUT/cfg2_default_pmc_by_kernel.json" "$OUT/pmc_cfg2_a" "$OUT/pmc_cfg2_b" "$OUT/pmc_cfg2_w" "$OUT/pmc_cfg2_f" --match tile_
  pmcrun ext_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --workload exterior --steps 20 --warmup 5
  pmcrun ext_w "WRITE_SIZE" --workload exterior --steps 20 --warmup 5; pmcrun ext_f "FETCH_SIZE" --workload exterior --steps 20 --warmup 5
  python scripts/pmc_summary.py "$OUT/exterior_default_pmc_by_kernel.json" "$OUT/pmc_ext_a" "$OUT/pmc_ext_w" "$OUT/pmc_ext_f" --match tile_
  rm -rf "$OUT"/pmc_ext_?
  for K in group refill; do
    pmcrun cfg3_${K}_a "$C1" --workload cfg3 --kernel $K --steps 4 --warmup 1 --opt cycle_detect=0
    pmcrun cfg3_${K}_b "$C2" --workload cfg3 --kernel $K --steps 4 --warmup 1 --opt cycle_detect=0
    python scripts/pmc_summary.py "$OUT/cfg3_${K}_pmc.json" "$OUT/pmc_cfg3_${K}_a" "$OUT/pmc_cfg3_${K}_b" --match tile_
  done
  rm -rf "$OUT"/pmc_cfg2_? "$OUT"/pmc_cfg3_*_?
fi
if ! skip power; then echo "== power traces"
  for spec in "cfg3 group 150" "cfg3 refill 150" "inset default 600" "cfg2 default 4000"; do set -- $spec
    timeout 300 python scripts/power_trace.py "$OUT/power_$1_$2.json" -- python bench.py --workload $1 --kernel $2 --no-cpu-baseline --no-extras --steps $3 --opt cycle_detect=0 > "$OUT/power_$1_$2.log" 2>&1
    python - "$OUT/power_$1_$2.json" <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); print("  ", r["bench"]["workload"][:8], r["bench"]["kernel"], "cap", r.get("power_cap_W"), "busy W p50", r.get("busy_power_W",{}).get("p50"), "sclk p50", r.get("busy_sclk_MHz",{}).get("p50"), "J/Gpi", round(r.get("J_per_G_pixel_iteration",0),4), "G/s", round(r["bench"]["value"],1))
except Exception as e: print("  power FAILED", e)
PY
  done
fi
if ! skip e2e; then echo "== level rate / worker end to end"; timeout 200 python scripts/level_rate.py 16 1024 > "$OUT/level16.log" 2>&1; grep "level\|two" "$OUT/level16.log"
  timeout 900 python scripts/worker_e2e.py 12 256 3 > "$OUT/worker_e2e.log" 2>&1; grep -v amdgpu.ids "$OUT/worker_e2e.log" | tail -14; fi
du -sh "$OUT"
