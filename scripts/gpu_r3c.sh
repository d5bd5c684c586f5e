#!/bin/bash
# Round-3 third GPU pass: deferred replay, same-box A/B against the previous loops (.ab/prev), parity first.
set -u
TAG=${1:-r3c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
echo "== parity (new loops): focused tests"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "golden or seeded or ragged or mrd_edge or cycle or f32 or smooth or fuzz or option_matrix or full_size_cfg2 or tiny_imag or prepass" > "$OUT/pytest_focus.log" 2>&1; echo "pytest(focus) rc=$?"; tail -3 "$OUT/pytest_focus.log"; grep -E "^FAILED|^ERROR" "$OUT/pytest_focus.log" | cut -c1-220 | head
echo "== soak"; timeout 300 python scripts/gpu_soak.py 90 31 > "$OUT/soak.log" 2>&1; tail -2 "$OUT/soak.log"
pm() { name=$1; shift; pmcrun $name "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU" "$@"; python scripts/pmc_summary.py "$OUT/${name}_pmc.json" "$OUT/pmc_$name" --match tile_; rm -rf "$OUT/pmc_$name"; }
runset() { tag=$1
  b ${tag}_cfg2 --no-cpu-baseline --no-extras
  b ${tag}_cfg3 --workload cfg3 --no-cpu-baseline --no-extras
  b ${tag}_chunk_l1 --workload chunk_l1 --no-cpu-baseline --no-extras
  b ${tag}_cfg2_f32 --precision f32 --no-cpu-baseline --no-extras
  b ${tag}_cfg5 --workload cfg5 --no-cpu-baseline --no-extras
  pm ${tag}_cfg2 --steps 10 --warmup 3
  pm ${tag}_cfg3 --workload cfg3 --steps 3 --warmup 1; }
echo "== new loops"; runset new
echo "== previous loops (same box): rebuild with .ab/prev/mbk_loops.inc"
cp distributedmandelbrot_amd/csrc/mbk_loops.inc /tmp/mbk_loops_new.inc
cp .ab/prev/mbk_loops.inc distributedmandelbrot_amd/csrc/mbk_loops.inc
python -m distributedmandelbrot_amd.build --force > "$OUT/build_prev.log" 2>&1; echo "build prev rc=$?"
runset prev
cp /tmp/mbk_loops_new.inc distributedmandelbrot_amd/csrc/mbk_loops.inc
python -m distributedmandelbrot_amd.build --force > "$OUT/build_new.log" 2>&1; echo "build new rc=$?"
b new_cfg2_again --no-cpu-baseline --no-extras
echo "== worker end to end (server with buffer pool)"
timeout 600 python scripts/worker_e2e.py 12 256 3 > "$OUT/worker_e2e.log" 2>&1; grep -v amdgpu.ids "$OUT/worker_e2e.log"
du -sh "$OUT"
