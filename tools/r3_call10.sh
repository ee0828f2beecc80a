#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03m
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for W in c2 c3 c5 c1; do python bench.py --workload $W --no-cpu-baseline --no-full-run --no-stateless --profile-all > gpurun_out/r03m/$W.json 2> gpurun_out/r03m/${W}_breakdown.txt; python -c "
import json; d=json.load(open('gpurun_out/r03m/$W.json')); print('$W', round(d['ms_per_step'],3))"; grep "x2h_k\|x2h_v" gpurun_out/r03m/${W}_breakdown.txt; done
python tools/wg_balance.py --detail 2>&1 | grep -v "^/opt" > gpurun_out/r03m/wg_balance_detail.txt; grep "^  value\|^  key" gpurun_out/r03m/wg_balance_detail.txt
