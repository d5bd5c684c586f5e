#!/bin/bash
# Run one pytest selection N times, uncaptured, and stop at the first failure (hunting a flaky crash).  scripts/repeat_test.sh N OUT -k EXPR [pytest args]
N=$1; OUT=$2; shift 2
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for i in $(seq 1 "$N"); do
  echo "=== run $i" >> "$OUT"
  timeout 600 python -m pytest tests -m gpu -x -q -s -p no:cacheprovider "$@" >> "$OUT" 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "run $i FAILED rc=$rc" | tee -a "$OUT"; tail -60 "$OUT"; exit 1; fi
done
echo "all $N runs passed" | tee -a "$OUT"
