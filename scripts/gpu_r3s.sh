#!/bin/bash
# r3s: the driver's plain command on the committed tree (profiles refreshed: `traffic` must be quoted again)
set -u
TAG=${1:-r3s}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
source scripts/gpu_lib.sh
b cfg2_default
python - "$OUT/bench_cfg2_default.log" <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(json.dumps(r["roofline"])[:900]); print({k:r[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","dtype","scaling","vs_baseline")})
PY
