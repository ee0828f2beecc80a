#!/bin/bash
# Round-3 GPU call 2: chunk-loop edge kernels on general graphs (stateless), node_proj variants.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03b
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_graph_modes.py tests/test_gpu_parity.py -m gpu -x -q > "$OUT/pytest_gpu.txt" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.txt"
tail -15 "$OUT/pytest_gpu.txt"
for V in "node_proj_bpipe=0" "node_proj_bpipe=1"; do
  timeout 300 python bench.py --no-cpu-baseline --no-full-run --no-stateless --profile-all --option $V > "$OUT/bench_c2_$V.json" 2> "$OUT/bench_c2_${V}_breakdown.txt"
  python -c "import json,sys; d=json.load(open('$OUT/bench_c2_$V.json')); print('$V', d['ms_per_step'])"; grep node "$OUT/bench_c2_${V}_breakdown.txt"
done
for G in "--knn 48" "--knn 64" "--cutoff-mode hybrid" "--cutoff-mode radius --radius 6.0 --cap 48"; do
  T=$(echo $G | tr -d ' -' | tr '.' '_')
  timeout 300 python bench.py --workload c5 --no-cpu-baseline --no-stateless --profile-all $G > "$OUT/bench_c5_$T.json" 2> "$OUT/bench_c5_${T}_breakdown.txt"
  python -c "import json,sys; d=json.load(open('$OUT/bench_c5_$T.json')); print('$T', d['ms_per_step'])"; grep -E "x2h|h2x|node|knn|gate" "$OUT/bench_c5_${T}_breakdown.txt"
done
