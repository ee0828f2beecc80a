#!/bin/bash
# rocprofv3 kernel statistics of a C5-size step on a `hybrid` and on a k = 48 graph (through gpurun):
#   tools/prof_general_graphs.sh [out dir under the repo, default gpurun_out/general]  ->  <out>/c5_{hybrid,knn48}_kernel_stats.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-gpurun_out/general}; mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
for M in hybrid knn48; do
  case $M in hybrid) F="--cutoff-mode hybrid";; knn48) F="--knn 48";; esac
  rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/stats_$M" -o $M -- python "$ROOT/bench.py" --workload c5 $F --no-cpu-baseline --no-full-run --no-stateless --no-sweep > "$ROOT/$OUT/bench_c5_${M}_under_rocprof.json" 2> "$ROOT/$OUT/stats_$M.log"
  find "$ROOT/$OUT/stats_$M" -name "*.db" | head -1 | xargs -r python "$ROOT/tools/rocprof_summary.py" > "$ROOT/$OUT/c5_${M}_kernel_stats.txt" 2>> "$ROOT/$OUT/stats_$M.log"
  rm -rf "$ROOT/$OUT/stats_$M"
done
