#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03q
TD_H2X12=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for W in c2 c1 c5; do for O in new old new old; do
  if [ $O = new ]; then export TD_H2X12=1; else unset TD_H2X12; fi
  timeout 300 python bench.py --workload $W --no-cpu-baseline --no-full-run --no-stateless --profile-all > gpurun_out/r03q/${W}_$O.json 2> gpurun_out/r03q/${W}_${O}_breakdown.txt; python -c "
import json; d=json.load(open('gpurun_out/r03q/${W}_$O.json')); print('$W $O', round(d['ms_per_step'],3))"; grep "h2x_k" gpurun_out/r03q/${W}_${O}_breakdown.txt | tr '\n' ' '; echo; done; done
TD_H2X12=1 python tools/wg_balance.py 2>&1 | grep "^  h2x     [345]" | cut -c1-130
