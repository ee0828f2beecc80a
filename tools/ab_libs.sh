#!/bin/bash
# A/B of two builds of the library inside one gpurun call: targetdiff_amd/lib/variant_A.so (before) / variant_B.so (after)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/ab
WL=${WL:-"c2 c3 c5 c1"}
for W in $WL; do for V in A B A B; do
  cp targetdiff_amd/lib/variant_$V.so targetdiff_amd/lib/libtargetdiff_hip.so
  timeout 300 python bench.py --workload $W --no-cpu-baseline --no-full-run --no-stateless --no-sweep --profile-all $EXTRA > gpurun_out/ab/${W}_$V.json 2> gpurun_out/ab/${W}_${V}_breakdown.txt; python -c "
import json; d=json.load(open('gpurun_out/ab/${W}_$V.json')); print('$W $V', round(d['ms_per_step'],3))"; grep "x2h_k\|x2h_v\|node_proj\|h2x_k" gpurun_out/ab/${W}_${V}_breakdown.txt | awk '{printf "%s %s  ", $1, $2}'; echo; done; done
cp targetdiff_amd/lib/variant_B.so targetdiff_amd/lib/libtargetdiff_hip.so
[ -n "$TESTS" ] && timeout 600 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -3
