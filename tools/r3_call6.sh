#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03f
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.txt" 2>&1; tail -3 "$OUT/pytest_gpu.txt"
for V in 0 1; do
  timeout 300 python bench.py --no-cpu-baseline --no-full-run --no-stateless --profile-all --option edge_row_dealing=$V > "$OUT/bench_c2_deal$V.json" 2> "$OUT/bench_c2_deal${V}_breakdown.txt"
  python -c "import json,sys; d=json.load(open('$OUT/bench_c2_deal$V.json')); print('deal=$V', d['ms_per_step'])"; grep -E "x2h|node" "$OUT/bench_c2_deal${V}_breakdown.txt"
done
timeout 300 python tools/wg_balance.py > "$OUT/wg_balance_c2.txt" 2>/dev/null; cat "$OUT/wg_balance_c2.txt"
for W in c3 c5; do
for V in 0 1; do
  timeout 300 python bench.py --workload $W --no-cpu-baseline --no-stateless --option edge_row_dealing=$V > "$OUT/bench_${W}_deal$V.json" 2>/dev/null
  python -c "import json,sys; d=json.load(open('$OUT/bench_${W}_deal$V.json')); print('$W deal=$V', d['ms_per_step'])"
done; done
