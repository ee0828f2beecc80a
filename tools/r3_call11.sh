#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03n
for W in c2 c3 c5; do for O in 2 3 2 3; do python bench.py --workload $W --no-cpu-baseline --no-full-run --no-stateless --profile-all --option edge_row_dealing=$O > gpurun_out/r03n/${W}_deal$O.json 2> gpurun_out/r03n/${W}_deal${O}_breakdown.txt; python -c "
import json; d=json.load(open('gpurun_out/r03n/${W}_deal$O.json')); print('$W deal=$O', round(d['ms_per_step'],3))"; grep "x2h_k\|x2h_v" gpurun_out/r03n/${W}_deal${O}_breakdown.txt | tr '\n' ' '; echo; done; done
python tools/wg_balance.py --option edge_row_dealing=3 2>&1 | grep "^  key\|^  value"
