#!/bin/bash
# quick same-box A/B (working tree vs .ab/prev_csrc) on cfg2 / chunk_l1 / cfg3 + a kernel trace of each build
set -u
TAG=${1:-ab2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
CS=distributedmandelbrot_amd/csrc
cp -r $CS /tmp/csrc_new
use() { rm -rf $CS; cp -r "$1" $CS; python -m distributedmandelbrot_amd.build --force > "$OUT/build_$2.log" 2>&1; echo "build $2 rc=$?"; }
runset() { tag=$1
  b ${tag}_cfg2 --no-cpu-baseline --no-extras
  b ${tag}_chunk_l1 --workload chunk_l1 --no-cpu-baseline --no-extras
  b ${tag}_cfg3 --workload cfg3 --no-cpu-baseline --no-extras --steps 10; }
for rep in 1 2; do
  echo "== new ($rep)"; [ $rep = 1 ] || use /tmp/csrc_new new; runset new$rep
  [ $rep = 1 ] && trace new_cfg2 --no-extras
  echo "== prev ($rep)"; use .ab/prev_csrc prev; runset prev$rep
done
use /tmp/csrc_new new
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "prepass or golden_full or launches_on_many" > "$OUT/pytest_focus.log" 2>&1; echo "pytest(focus) rc=$?"; tail -2 "$OUT/pytest_focus.log"
