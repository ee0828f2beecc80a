#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03d
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -q -s > "$OUT/pytest_gpu.txt" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.txt"
grep -E "1h36 x 2|teacher|passed|failed|FAILED|forward_c5|hybrid 20" "$OUT/pytest_gpu.txt" | tail -25
for G in "--knn 48" "--knn 64" "--cutoff-mode hybrid"; do
  T=$(echo $G | tr -d ' -' | tr '.' '_')
  timeout 300 python bench.py --workload c5 --no-cpu-baseline --no-stateless --profile-all $G > "$OUT/bench_c5_$T.json" 2> "$OUT/bench_c5_${T}_breakdown.txt"
  python -c "import json,sys; d=json.load(open('$OUT/bench_c5_$T.json')); print('$T', d['ms_per_step'])"; grep -E "x2h|h2x" "$OUT/bench_c5_${T}_breakdown.txt"
done
