#!/bin/bash
# C5 sweep A/B of every targetdiff_amd/lib/variant_*.so inside one gpurun call:  MODES="k48 k64 hybrid k32" tools/ab_c5.sh
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/ab
MODES=${MODES:-"k32 k48 k64 hybrid"}; ROUNDS=${ROUNDS:-1}
VARS=$(ls targetdiff_amd/lib/variant_*.so | sed 's/.*variant_\(.*\)\.so/\1/')
for M in $MODES; do
  case $M in k32) F="--knn 32";; k48) F="--knn 48";; k64) F="--knn 64";; hybrid) F="--cutoff-mode hybrid";; esac
  for R in $(seq $ROUNDS); do for V in $VARS; do
    cp targetdiff_amd/lib/variant_$V.so targetdiff_amd/lib/libtargetdiff_hip.so
    timeout 300 python bench.py --workload c5 $F --no-cpu-baseline --no-stateless --profile-all $EXTRA > gpurun_out/ab/c5_${M}_${V}_$R.json 2> gpurun_out/ab/c5_${M}_${V}_${R}_breakdown.txt
    python -c "
import json; d=json.load(open('gpurun_out/ab/c5_${M}_${V}_$R.json')); print('c5 $M $V $R', round(d['ms_per_step'],3))"
    grep "x2h_k\|x2h_v\|node_proj\|h2x_k" gpurun_out/ab/c5_${M}_${V}_${R}_breakdown.txt | awk '{printf "%s %s  ", $1, $2}'; echo
  done; done; done
LAST=$(echo $VARS | awk '{print $NF}')
cp targetdiff_amd/lib/variant_$LAST.so targetdiff_amd/lib/libtargetdiff_hip.so
