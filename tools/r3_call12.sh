#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03o
python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph_modes.py -m gpu -x -q 2>&1 | tail -3
for W in c2 c1; do for O in new old new old; do
  if [ $O = old ]; then export TD_NP_NOTRANS=1; else unset TD_NP_NOTRANS; fi
  python bench.py --workload $W --no-cpu-baseline --no-full-run --no-stateless --profile-all > gpurun_out/r03o/${W}_$O.json 2> gpurun_out/r03o/${W}_${O}_breakdown.txt; python -c "
import json; d=json.load(open('gpurun_out/r03o/${W}_$O.json')); print('$W $O', round(d['ms_per_step'],3))"; grep "node_proj\|x2h_k" gpurun_out/r03o/${W}_${O}_breakdown.txt | tr '\n' ' '; echo; done; done
unset TD_NP_NOTRANS
python bench.py --workload c3 --no-cpu-baseline --no-stateless --profile-all 2>&1 >/dev/null | grep "node_proj"
TD_NP_NOTRANS=1 python bench.py --workload c3 --no-cpu-baseline --no-stateless --profile-all 2>&1 >/dev/null | grep "node_proj"
