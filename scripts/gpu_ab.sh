#!/bin/bash
# Same-box A/B of the working tree's kernels against the csrc directory stashed in .ab/prev_csrc (the committed
# state): bench lines for a few workloads with each build, twice, interleaved.  Usage: scripts/gpu_ab.sh TAG
set -u
TAG=${1:-ab}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
CS=distributedmandelbrot_amd/csrc
cp -r $CS /tmp/csrc_new
use() { rm -rf $CS; cp -r "$1" $CS; python -m distributedmandelbrot_amd.build --force > "$OUT/build_$2.log" 2>&1; echo "build $2 rc=$?"; }
runset() { tag=$1
  b ${tag}_cfg2 --no-cpu-baseline --no-extras
  b ${tag}_chunk_l1 --workload chunk_l1 --no-cpu-baseline --no-extras
  b ${tag}_cfg3 --workload cfg3 --no-cpu-baseline --no-extras --steps 10
  b ${tag}_exterior_group --workload exterior --kernel group --no-cpu-baseline --no-extras
  b ${tag}_cfg1 --workload cfg1 --no-cpu-baseline --no-extras; }
echo "== parity of the new build (focused)"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "golden or seeded or ragged or option_matrix or full_size_cfg2 or prepass" > "$OUT/pytest_focus.log" 2>&1; echo "pytest(focus) rc=$?"; tail -2 "$OUT/pytest_focus.log"
for rep in 1 2; do
  echo "== new ($rep)"; [ $rep = 1 ] || use /tmp/csrc_new new; runset new$rep
  echo "== prev ($rep)"; use .ab/prev_csrc prev; runset prev$rep
done
use /tmp/csrc_new new
du -sh "$OUT"
