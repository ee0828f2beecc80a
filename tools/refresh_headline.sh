#!/bin/bash
# Headline evidence of the current build in ONE gpurun call, all on the same library file: PMC passes (counters only + kernel trace:
# SQ instruction / busy counters, FETCH_SIZE, WRITE_SIZE in separate passes) -> profiles/traffic_x2h_*.json stamped with the library's
# build tag; kernel stats of the default command (rocprofv3 --kernel-trace --stats); the default bench line, which reports
# `roofline.traffic` only when its own build tag equals the traffic files' -- the script FAILS if they differ; per-class breakdown,
# workgroup traces, C1 / C3 / C5 / k = 48 lines.
#   usage: tools/refresh_headline.sh <tag>  -> gpurun_out/<tag>/   (copy what is to be kept into profiles/<tag>_*)
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:-headline}
OUT=gpurun_out/$TAG; mkdir -p $OUT
ROOT=$(pwd)
PMC_SHORT=1 bash tools/pmc_collect.sh $OUT/pmc_c2 > $OUT/pmc_c2.log 2>&1
python tools/pmc_summary.py $OUT/pmc_c2 > $OUT/pmc_c2.txt
cp $OUT/pmc_c2.txt profiles/${TAG}_pmc_c2.txt
python tools/traffic_from_pmc.py profiles/${TAG}_pmc_c2.txt c2 > $OUT/traffic.log 2>&1
cp profiles/traffic_x2h_value.json profiles/traffic_x2h_key.json $OUT/ 2>/dev/null
rm -rf $OUT/pmc_c2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/stats_c2" -o c2 -- python "$ROOT/bench.py" --workload c2 --no-cpu-baseline --no-full-run --no-stateless --no-sweep > "$ROOT/$OUT/bench_c2_under_rocprof.json" 2> "$ROOT/$OUT/stats_c2.log"
find "$ROOT/$OUT/stats_c2" -name "*.db" | head -1 | xargs -r python "$ROOT/tools/rocprof_summary.py" > "$ROOT/$OUT/c2_kernel_stats.txt" 2>> "$ROOT/$OUT/stats_c2.log"
rm -rf "$ROOT/$OUT/stats_c2"
cd "$ROOT"
python bench.py --no-cpu-baseline --profile-all --no-full-run --no-sweep > $OUT/bench_c2_profile_all.json 2> $OUT/bench_c2_breakdown.txt
python tools/wg_balance.py > $OUT/wg_balance_c2.txt 2>/dev/null
# the same counters at C3 (BASELINE's HBM-roofline run) and C5, so that their bench lines carry roofline.traffic of this build too
for W in c3 c5; do
  BENCH_ARGS="--workload $W" PMC_SHORT=1 bash tools/pmc_collect.sh $OUT/pmc_$W > $OUT/pmc_$W.log 2>&1
  python tools/pmc_summary.py $OUT/pmc_$W > $OUT/pmc_$W.txt
  cp $OUT/pmc_$W.txt profiles/${TAG}_pmc_$W.txt
  python tools/traffic_from_pmc.py profiles/${TAG}_pmc_$W.txt $W >> $OUT/traffic.log 2>&1
  cp profiles/traffic_x2h_value_$W.json profiles/traffic_x2h_key_$W.json $OUT/ 2>/dev/null
  rm -rf $OUT/pmc_$W
done
for W in c1 c3 c5; do python bench.py --workload $W --no-cpu-baseline > $OUT/bench_$W.json 2>/dev/null; done
python bench.py --workload c5 --no-cpu-baseline --knn 48 > $OUT/bench_c5_knn48.json 2>/dev/null
python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python - <<PY || exit 1
import json, sys
sys.path.insert(0, '.')
from targetdiff_amd import capi
tag = capi.build_tag()
for n in ('bench_c2','bench_c1','bench_c3','bench_c5','bench_c5_knn48'):
    d=json.load(open('$OUT/%s.json' % n)); r=d['roofline']; print(n, round(d['ms_per_step'],3), round(d['value'],2), round(r['frac'],3), round(r['key_pass']['frac'],3), round(r['whole_step']['executed_frac_of_fp32_peak'],3), d.get('stateless_ms_per_step'))
d=json.load(open('$OUT/bench_c2.json'))
print('full run', d['full_run']['wall_s'], d['full_run']['ligands_per_s'])
print('sweep', {k: round(v['ms_per_step'], 3) for k, v in d['geometry_sweep'].items() if isinstance(v, dict)})
for f in ('traffic_x2h_value.json', 'traffic_x2h_key.json'):
    t = json.load(open('profiles/' + f))
    if t.get('build_tag') != tag or d['config']['build_tag'] != tag:
        sys.exit(f'{f}: build tag {t.get("build_tag")} != library {tag} / bench {d["config"]["build_tag"]}')
if d['roofline']['traffic'] is None:
    sys.exit('bench line carries no roofline.traffic although the PMC passes ran on this build')
print('build tag', tag, 'traffic per value launch', d['roofline']['traffic'], 'B =', round(d['roofline']['hbm_frac'], 3), 'of the HBM peak')
PY
