#!/bin/bash
# Round-3 GPU call 1: the GPU suite on the current build, node_proj A/B, the default bench line, HBM counters for C3.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03a
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.txt" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.txt"
tail -5 "$OUT/pytest_gpu.txt"
for A in 0 1; do
  timeout 300 python bench.py --no-cpu-baseline --no-full-run --no-stateless --profile-all --option node_proj_async=$A > "$OUT/bench_c2_async$A.json" 2> "$OUT/bench_c2_async${A}_breakdown.txt"
  tail -c 300 "$OUT/bench_c2_async$A.json"; grep node "$OUT/bench_c2_async${A}_breakdown.txt"
done
timeout 600 python bench.py > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"; tail -c 1500 "$OUT/bench_c2.json"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_c2" -o c2 -- python "$ROOT/bench.py" --no-cpu-baseline --no-full-run --no-stateless > "$OUT/bench_c2_under_rocprof.json" 2> "$OUT/stats_c2.log"
find "$OUT/stats_c2" -name "*.db" | head -1 | xargs -r python "$ROOT/tools/rocprof_summary.py" > "$OUT/c2_kernel_stats.txt" 2>> "$OUT/stats_c2.log"
rm -rf "$OUT/stats_c2"; head -12 "$OUT/c2_kernel_stats.txt"
# HBM counters on C3 (BASELINE's "HBM roofline run"): FETCH_SIZE and WRITE_SIZE in separate passes
for P in "tcc1 FETCH_SIZE GRBM_GUI_ACTIVE" "tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  set -- $P; NAME=$1; shift
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_c3/$NAME" -o p -- \
      python "$ROOT/bench.py" --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-stateless > "$OUT/pmc_c3_$NAME.log" 2>&1
done
python "$ROOT/tools/pmc_summary.py" "$OUT/pmc_c3" > "$OUT/pmc_c3_summary.txt" 2>> "$OUT/pmc_c3.log"
find "$OUT/pmc_c3" -name "*.csv" -size +2M -delete
head -30 "$OUT/pmc_c3_summary.txt"
