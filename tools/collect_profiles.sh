#!/bin/bash
# Round-end evidence run on the GPU box: kernel stats (rocprofv3 --kernel-trace --stats), PMC passes, bench JSON lines.
# usage (from the repo root, through gpurun): tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...
set -u
TAG=${1:-final}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o c2 -- python "$ROOT/bench.py" --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
find "$OUT/stats" -name "*.db" | head -1 | xargs -r python "$ROOT/tools/rocprof_summary.py" > "$OUT/kernel_stats.txt" 2>> "$OUT/stats.log"
cd "$ROOT" && bash tools/pmc_collect.sh "gpurun_out/$TAG/pmc" > "$OUT/pmc.log" 2>&1
python tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary.txt" 2>> "$OUT/pmc.log"
find "$OUT/pmc" -name "*.csv" -size +2M -delete
python bench.py --no-cpu-baseline --profile-all > "$OUT/bench_profile_all.json" 2> "$OUT/bench_breakdown.txt"
python bench.py > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
python bench.py --no-cpu-baseline --initial-state > "$OUT/bench_c2_initial_state.json" 2>> "$OUT/bench_c2.err"
tail -c 600 "$OUT/bench_c2.json"
