#!/bin/bash
# Round-3 second GPU pass: the new tests, prepass-overlap A/B, light-tile changes, worker end to end (Python vs native
# feeder), cfg3 counters (group vs refill), kernel trace + PMC of the headline.  Usage: scripts/gpu_r3b.sh TAG
set -u
TAG=${1:-r3b}; SKIP=" ${SKIP:-} "
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
skip() { [[ "$SKIP" == *" $1 "* ]]; }
source scripts/gpu_lib.sh
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12 > "$OUT/rocminfo.txt" 2>&1; nproc > "$OUT/nproc.txt"
echo "== new tests first"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "slot0 or cfg3_as or cfg3_data or power_of_two or prepass or native_worker or history or pci_bus or any_arrival or lazy_uniform" > "$OUT/pytest_new.log" 2>&1; echo "pytest(new) rc=$?"; tail -3 "$OUT/pytest_new.log"; grep -E "^FAILED|^ERROR" "$OUT/pytest_new.log" | cut -c1-220 | head
echo "== bench: prepass overlap A/B"
b cfg2_default
b cfg2_serial_prepass --opt prepass_overlap=0 --no-cpu-baseline --no-extras
b cfg2_default_b --no-cpu-baseline --no-extras
b cfg2_serial_prepass_b --opt prepass_overlap=0 --no-cpu-baseline --no-extras
b cfg2_probe16 --opt probe_steps=16 --no-cpu-baseline --no-extras
b cfg2_probe64 --opt probe_steps=64 --no-cpu-baseline --no-extras
b cfg2_streams2 --streams 2 --no-cpu-baseline --no-extras
echo "== light tiles"
b exterior --workload exterior --no-cpu-baseline; b exterior_both --workload exterior --outputs both --no-cpu-baseline
b chunk_l1 --workload chunk_l1 --no-cpu-baseline; b cfg1 --workload cfg1 --no-cpu-baseline
b cfg3 --workload cfg3 --no-cpu-baseline; b cfg3_refill --workload cfg3 --kernel refill --no-cpu-baseline
if ! skip e2e; then echo "== level rate / worker end to end"; timeout 200 python scripts/level_rate.py 16 1024 > "$OUT/level16.log" 2>&1; grep "level\|two" "$OUT/level16.log"
  timeout 600 python scripts/worker_e2e.py 12 256 3 > "$OUT/worker_e2e.log" 2>&1; grep -v amdgpu.ids "$OUT/worker_e2e.log"; fi
echo "== rocprofv3 kernel traces"
trace cfg2_default --no-extras
trace exterior_default --workload exterior
if ! skip pmc; then echo "== rocprofv3 pmc: cfg3 group vs refill (one pass each), cfg2 headline"
  pmcrun cfg3_group "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --workload cfg3 --kernel group --steps 4 --warmup 1 --opt cycle_detect=0
  pmcrun cfg3_refill "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --workload cfg3 --kernel refill --steps 4 --warmup 1 --opt cycle_detect=0
  python scripts/pmc_summary.py "$OUT/cfg3_group_vs_refill_pmc.json" "$OUT/pmc_cfg3_group" "$OUT/pmc_cfg3_refill" --match tile_
  head -2 "$(find "$OUT/pmc_cfg3_group" -name '*counter_collection.csv' | head -1)" > "$OUT/pmc_csv_header.txt"
  pmcrun cfg2_a "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" --steps 20 --warmup 5
  pmcrun cfg2_b "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH" --steps 20 --warmup 5
  pmcrun cfg2_w "WRITE_SIZE" --steps 20 --warmup 5; pmcrun cfg2_f "FETCH_SIZE" --steps 20 --warmup 5
  python scripts/pmc_summary.py "$OUT/cfg2_default_pmc_by_kernel.json" "$OUT/pmc_cfg2_a" "$OUT/pmc_cfg2_b" "$OUT/pmc_cfg2_w" "$OUT/pmc_cfg2_f" --match tile_
  rm -rf "$OUT"/pmc_cfg3_group "$OUT"/pmc_cfg3_refill "$OUT"/pmc_cfg2_a "$OUT"/pmc_cfg2_b "$OUT"/pmc_cfg2_w "$OUT"/pmc_cfg2_f
fi
if ! skip tests; then echo "== pytest gpu (full)"; timeout 2400 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"; grep -E "^FAILED|^ERROR" "$OUT/pytest_gpu.log" | cut -c1-220 | head -10; fi
python -c "
import sys; sys.path.insert(0, '.'); import bench; print('kernel source sha256', bench.kernel_source_hash())" | tee "$OUT/source_sha256.txt"
du -sh "$OUT"
