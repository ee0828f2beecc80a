#!/bin/bash
# Round-3 evidence run on the GPU box: kernel stats (rocprofv3 --kernel-trace --stats) for BASELINE configs 1, 2, 3, 5, the C5
# fan-in / fan-out-cap sweep, PMC passes for C2 (all counter groups) and C3 / C5 (HBM traffic), the step timeline, the
# workgroup balance, the small-batch driver, the default bench line.
# usage (from the repo root, through gpurun): tools/collect_profiles_r3.sh <tag> [quick]     -> gpurun_out/<tag>/...
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for W in c2 c3 c5 c1; do
  rocprofv3 --kernel-trace --stats -d "$OUT/stats_$W" -o $W -- python "$ROOT/bench.py" --workload $W --no-cpu-baseline --no-full-run --no-stateless \
      > "$OUT/bench_${W}_under_rocprof.json" 2> "$OUT/stats_$W.log"
  find "$OUT/stats_$W" -name "*.db" | head -1 | xargs -r python "$ROOT/tools/rocprof_summary.py" > "$OUT/${W}_kernel_stats.txt" 2>> "$OUT/stats_$W.log"
  rm -rf "$OUT/stats_$W"
done
cd "$ROOT"
for W in c1 c3 c5; do python bench.py --workload $W --no-cpu-baseline --profile-all > "$OUT/bench_$W.json" 2> "$OUT/bench_${W}_breakdown.txt"; done
# C5 sweep: k-NN fan-in and radius fan-out cap, hybrid
for K in 16 48 64; do python bench.py --workload c5 --no-cpu-baseline --knn $K --profile-all > "$OUT/bench_c5_knn$K.json" 2> "$OUT/bench_c5_knn${K}_breakdown.txt"; done
for C in 16 32 48 64; do python bench.py --workload c5 --no-cpu-baseline --no-stateless --cutoff-mode radius --radius 6.0 --cap $C > "$OUT/bench_c5_radius6_cap$C.json" 2>> "$OUT/sweep.err"; done
python bench.py --workload c5 --no-cpu-baseline --cutoff-mode hybrid --profile-all > "$OUT/bench_c5_hybrid.json" 2> "$OUT/bench_c5_hybrid_breakdown.txt"
python bench.py --workload c2 --no-cpu-baseline --no-full-run --cutoff-mode hybrid > "$OUT/bench_c2_hybrid.json" 2>> "$OUT/sweep.err"
python bench.py --workload c4 --no-cpu-baseline > "$OUT/bench_c4_1gpu.json" 2>> "$OUT/sweep.err"
python -m pytest tests/test_gpu_graph_modes.py -q -s -m gpu -k options > "$OUT/split_error_table.txt" 2>&1
python bench.py --no-cpu-baseline --profile-all --no-full-run > "$OUT/bench_c2_profile_all.json" 2> "$OUT/bench_c2_breakdown.txt"
python bench.py --no-cpu-baseline --no-full-run --initial-state > "$OUT/bench_c2_initial_state.json" 2>> "$OUT/sweep.err"
python tools/wg_balance.py > "$OUT/wg_balance_c2.txt" 2>> "$OUT/sweep.err"
python tools/small_batch_bench.py > "$OUT/small_batch_16.json" 2>> "$OUT/sweep.err"
python tools/full_run.py > "$OUT/full_run_c2.json" 2>> "$OUT/sweep.err"
if [ "${2:-}" != "quick" ]; then
  bash tools/pmc_collect.sh "gpurun_out/$TAG/pmc" > "$OUT/pmc.log" 2>&1
  python tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_c2.txt" 2>> "$OUT/pmc.log"
  find "$OUT/pmc" -name "*.csv" -size +2M -delete
  for W in c3 c5; do
    cd /tmp
    for P in "tcc1 FETCH_SIZE GRBM_GUI_ACTIVE" "tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
      set -- $P; NAME=$1; shift
      rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$W/$NAME" -o p -- \
          python "$ROOT/bench.py" --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-stateless > "$OUT/pmc_${W}_$NAME.log" 2>&1
    done
    cd "$ROOT"
    python tools/pmc_summary.py "$OUT/pmc_$W" > "$OUT/pmc_$W.txt" 2>> "$OUT/pmc.log"
    find "$OUT/pmc_$W" -name "*.csv" -size +2M -delete
  done
  python bench.py --cpu-full > "$OUT/bench_c2_cpu_full.json" 2> "$OUT/bench_c2_cpu_full.err"
fi
python bench.py > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
tail -c 600 "$OUT/bench_c2.json"
