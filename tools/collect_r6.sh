#!/bin/bash
# Round-6 evidence in one gpurun call on the final build: tools/refresh_headline.sh (PMC passes -> traffic files, rocprofv3 kernel stats, default
# bench line with roofline.traffic, breakdown, workgroup traces, C1 / C3 / C5 / k = 48 lines), the GPU suite's margins, the microbenchmarks
# of the round, the small-batch driver at the reference's batch sizes, the C4 set at BATCH_SIZE 50 / 100, the general-graph lines.
#   usage (through gpurun): tools/collect_r6.sh <tag>   -> gpurun_out/<tag>/
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:-r06}
OUT=gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | tail -60 > $OUT/gputests.txt
cp gpurun_out/margins.txt $OUT/test_margins.txt 2>/dev/null
tools/microbench/bin/valu_rates > $OUT/microbench_valu_rates.txt 2>&1
tools/microbench/bin/f16probe > $OUT/microbench_f16_split_probe.txt 2>&1
bash tools/refresh_headline.sh $TAG > $OUT/refresh.log 2>&1
tail -12 $OUT/refresh.log
python tools/small_batch_bench.py --samples 96 --batch-size 16 > $OUT/small_batch_16.json 2>/dev/null
python tools/small_batch_bench.py --samples 100 --batch-size 50 > $OUT/small_batch_50.json 2>/dev/null
python bench.py --workload c4 --batch-size 50 > $OUT/bench_c4_1gpu_b50.json 2>/dev/null
python bench.py --workload c4 > $OUT/bench_c4_1gpu.json 2>/dev/null
python bench.py --workload c5 --no-cpu-baseline --knn 64 > $OUT/bench_c5_knn64.json 2>/dev/null
python bench.py --workload c5 --no-cpu-baseline --cutoff-mode hybrid > $OUT/bench_c5_hybrid.json 2>/dev/null
python bench.py --no-cpu-baseline --no-full-run --no-sweep --no-stateless --option edge_second_layer_f16=0 > $OUT/bench_c2_fp32_second_layer.json 2>/dev/null
bash tools/prof_general_graphs.sh $OUT > /dev/null 2>&1; cd "${GRAFT_REPO_ROOT:-.}"
python - <<PY
import json
for n in ('bench_c4_1gpu_b50', 'bench_c4_1gpu', 'bench_c5_knn64', 'bench_c5_hybrid', 'bench_c2_fp32_second_layer'):
    try:
        d = json.load(open('$OUT/%s.json' % n)); print(n, round(d['ms_per_step'], 3), round(d['value'], 2))
    except Exception as e:
        print(n, 'failed', e)
for n in (16, 50):
    d = json.load(open('$OUT/small_batch_%d.json' % n)); print('batch', n, {k: round(v['ligands_per_s_at_1000_steps'], 2) for k, v in d.items() if isinstance(v, dict)})
PY
tail -3 $OUT/gputests.txt
