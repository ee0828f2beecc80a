#!/bin/bash
# A/B of the per-edge-block skipping + workgroup balance detail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03k
python -m pytest tests/test_gpu_graph_modes.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for K in 16 48 64; do python bench.py --workload c5 --no-cpu-baseline --no-stateless --knn $K --profile-all > gpurun_out/r03k/c5_knn$K.json 2> gpurun_out/r03k/c5_knn${K}_breakdown.txt; done
python bench.py --workload c5 --no-cpu-baseline --no-stateless --cutoff-mode hybrid > gpurun_out/r03k/c5_hybrid.json 2>/dev/null
python tools/wg_balance.py --detail > gpurun_out/r03k/wg_balance_detail.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03k/*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], round(d['ms_per_step'],3))
PY
