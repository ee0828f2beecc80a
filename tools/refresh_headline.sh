#!/bin/bash
# Headline evidence of the current build in one short gpurun call: kernel stats of the default command (rocprofv3), the default bench line,
# per-class breakdown, workgroup traces, C1 / C3 / C5 / k = 48 lines.   usage: tools/refresh_headline.sh <tag>  -> gpurun_out/<tag>/
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/${1:-headline}; mkdir -p $OUT
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/stats_c2" -o c2 -- python "$ROOT/bench.py" --workload c2 --no-cpu-baseline --no-full-run --no-stateless > "$ROOT/$OUT/bench_c2_under_rocprof.json" 2> "$ROOT/$OUT/stats_c2.log"
find "$ROOT/$OUT/stats_c2" -name "*.db" | head -1 | xargs -r python "$ROOT/tools/rocprof_summary.py" > "$ROOT/$OUT/c2_kernel_stats.txt" 2>> "$ROOT/$OUT/stats_c2.log"
rm -rf "$ROOT/$OUT/stats_c2"
cd "$ROOT"
python bench.py --no-cpu-baseline --profile-all --no-full-run > $OUT/bench_c2_profile_all.json 2> $OUT/bench_c2_breakdown.txt
python tools/wg_balance.py > $OUT/wg_balance_c2.txt 2>/dev/null
for W in c1 c3 c5; do python bench.py --workload $W --no-cpu-baseline > $OUT/bench_$W.json 2>/dev/null; done
python bench.py --workload c5 --no-cpu-baseline --knn 48 > $OUT/bench_c5_knn48.json 2>/dev/null
python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python -c "
import json
for n in ('bench_c2','bench_c1','bench_c3','bench_c5','bench_c5_knn48'):
    d=json.load(open('$OUT/%s.json' % n)); r=d['roofline']; print(n, round(d['ms_per_step'],3), round(d['value'],2), round(r['frac'],3), round(r['key_pass']['frac'],3), round(r['whole_step']['executed_frac_of_fp32_peak'],3), d.get('stateless_ms_per_step'))
d=json.load(open('$OUT/bench_c2.json')); print(d['full_run']['wall_s'], d['full_run']['ligands_per_s'])
"
