#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03p
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph_modes.py -m gpu -x -q 2>&1 | tail -3
for W in c2 c1 c5; do for O in 2 1 2 1; do
  timeout 300 python bench.py --workload $W --no-cpu-baseline --no-full-run --no-stateless --profile-all --option h2x_fused=$O > gpurun_out/r03p/${W}_$O.json 2> gpurun_out/r03p/${W}_${O}_breakdown.txt; python -c "
import json; d=json.load(open('gpurun_out/r03p/${W}_$O.json')); print('$W h2x_fused=$O', round(d['ms_per_step'],3))"; grep "h2x_k" gpurun_out/r03p/${W}_${O}_breakdown.txt | tr '\n' ' '; echo; done; done
