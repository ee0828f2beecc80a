#!/bin/bash
# Round-3 GPU call 3: caching session on general graphs; full GPU suite; C5 sweep
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03c
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.txt" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.txt"
tail -25 "$OUT/pytest_gpu.txt"
for G in "--knn 48" "--knn 64" "--cutoff-mode hybrid" "--knn 32"; do
  T=$(echo $G | tr -d ' -' | tr '.' '_')
  timeout 300 python bench.py --workload c5 --no-cpu-baseline --profile-all $G > "$OUT/bench_c5_$T.json" 2> "$OUT/bench_c5_${T}_breakdown.txt"
  python -c "import json,sys; d=json.load(open('$OUT/bench_c5_$T.json')); print('$T', d['ms_per_step'], d.get('stateless_ms_per_step'), d['roofline']['session_rows'])"; grep -E "x2h|h2x|node|knn|gate" "$OUT/bench_c5_${T}_breakdown.txt"
done
timeout 300 python bench.py --workload c2 --no-cpu-baseline --no-full-run --profile-all --cutoff-mode hybrid > "$OUT/bench_c2_hybrid.json" 2> "$OUT/bench_c2_hybrid_breakdown.txt"
python -c "import json,sys; d=json.load(open('$OUT/bench_c2_hybrid.json')); print('c2 hybrid', d['ms_per_step'], d.get('stateless_ms_per_step'))"
timeout 300 python bench.py --no-cpu-baseline --no-full-run --profile-all > "$OUT/bench_c2.json" 2> "$OUT/bench_c2_breakdown.txt"
python -c "import json,sys; d=json.load(open('$OUT/bench_c2.json')); print('c2', d['ms_per_step'], d.get('stateless_ms_per_step'))"; cat "$OUT/bench_c2_breakdown.txt"
