#!/bin/bash
# A/B/C... of several builds of the library inside ONE gpurun call (same box): every targetdiff_amd/lib/variant_*.so, alternated
# ROUNDS times per workload; the LAST variant (alphabetical) stays installed for the tests.
#   WL="c2 c1" ROUNDS=2 TESTS="tests/test_gpu_parity.py" EXTRA="--steps 30" tools/ab_variants.sh
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/ab
WL=${WL:-"c2 c1"}; ROUNDS=${ROUNDS:-2}
VARS=$(ls targetdiff_amd/lib/variant_*.so | sed 's/.*variant_\(.*\)\.so/\1/')
for W in $WL; do for R in $(seq $ROUNDS); do for V in $VARS; do
  cp targetdiff_amd/lib/variant_$V.so targetdiff_amd/lib/libtargetdiff_hip.so
  timeout 300 python bench.py --workload $W --no-cpu-baseline --no-full-run --no-stateless --no-sweep --profile-all $EXTRA > gpurun_out/ab/${W}_${V}_$R.json 2> gpurun_out/ab/${W}_${V}_${R}_breakdown.txt
  python -c "
import json; d=json.load(open('gpurun_out/ab/${W}_${V}_$R.json')); r=d['roofline']; print('$W $V $R', round(d['ms_per_step'],3), 'value', round(r['frac'],3), 'key', round(r['key_pass']['frac'],3))"
  grep "x2h_k\|x2h_v\|node_proj\|h2x_k\|gate\|knn" gpurun_out/ab/${W}_${V}_${R}_breakdown.txt | awk '{printf "%s %s  ", $1, $2}'; echo
done; done; done
LAST=$(echo $VARS | awk '{print $NF}')
cp targetdiff_amd/lib/variant_$LAST.so targetdiff_amd/lib/libtargetdiff_hip.so
[ -n "$TESTS" ] && timeout 900 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -5
