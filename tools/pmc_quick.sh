#!/bin/bash
# SQ counters + kernel durations of one library variant:  tools/pmc_quick.sh <variant> <outname> [kernel filter]
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd)
V=$1; OUT=gpurun_out/$2; mkdir -p $OUT
cp targetdiff_amd/lib/variant_$V.so targetdiff_amd/lib/libtargetdiff_hip.so
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -o p -- python "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-full-run --no-stateless --no-sweep ${BENCH_ARGS:-} > "$ROOT/$OUT/$name.log" 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run sq3 SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_LDS_UNALIGNED_STALL
cd "$ROOT"
python tools/pmc_summary.py $OUT > $OUT/summary.txt
python - <<PY > $OUT/durations.txt
import csv, glob, collections
d = collections.defaultdict(list)
for p in glob.glob('$OUT/sq1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        d[r['Kernel_Name'][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f'{k:60s} n={len(v):4d} total={sum(v):10.1f} avg={sum(v)/len(v):8.2f} max={max(v):8.2f}')
PY
grep -A 30 "${3:-node_proj}" $OUT/summary.txt | head -80; head -12 $OUT/durations.txt
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3
