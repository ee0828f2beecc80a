#!/bin/bash
# The whole multi-GPU table from ONE lease of an N-GPU node (no such hardware has been available to this project: every 2 / 4 / 8-GPU
# figure in README / DESIGN is a prediction until this has run).  Emits one bench line per (workload, N):
#   c2  weak scaling   -- every rank samples its own replica of BASELINE config 2 (1h36 x 100 samples)
#   c4  strong scaling -- the fixed 100-pocket x 100-sample job of scripts/batch_sample_diffusion.sh:15-20, pocket i -> rank i % N, at the
#       script's BATCH_SIZE=50 and as one 100-sample batch per pocket
# Every line carries the rank census (world size and backend as torch.distributed reports them; per rank: device ordinal, PCI address,
# UUID, host, pid, its own ms per step) and, for c4, load_balance.per_rank_seconds -- the measured counterpart of the predicted
# max / mean of tools/c4_per_pocket.py (round-robin 1.010 / 1.057 / 1.094 at 2 / 4 / 8 ranks).
#   tools/scale_sweep.sh [OUTDIR] [GPUS="1 2 4 8"] [STEPS=20]
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/scale}; GPUS=${2:-"1 2 4 8"}; STEPS=${3:-20}
mkdir -p "$OUT"
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
for N in $GPUS; do
  if [ "$N" -gt "$HAVE" ]; then echo "skip N=$N: $HAVE device(s) visible" | tee -a "$OUT/skipped.txt"; continue; fi
  run() {   # name, bench arguments
    local name=$1; shift
    if [ "$N" -eq 1 ]; then python bench.py --gpus 1 --steps "$STEPS" --warmup 3 "$@" > "$OUT/${name}_n$N.json" 2> "$OUT/${name}_n$N.err"
    else python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29500 + N)) \
           bench.py --gpus "$N" --steps "$STEPS" --warmup 3 "$@" > "$OUT/${name}_n$N.json" 2> "$OUT/${name}_n$N.err"; fi
    python - "$OUT/${name}_n$N.json" "$name" "$N" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    lb = d.get('load_balance', {}).get('max_over_mean')
    print(f"{sys.argv[2]:22s} N={sys.argv[3]}  {d['value']:8.2f} ligands/s  {d['ms_per_step']:8.3f} ms/step  world={d['ranks']['world_size']}"
          + (f"  max/mean={lb:.3f}" if lb else ''))
except Exception as e:
    print(f"{sys.argv[2]} N={sys.argv[3]}: no line ({e})")
PY
  }
  run c2_weak --workload c2 --no-cpu-baseline --no-full-run --no-sweep --no-stateless
  run c4_strong_b50 --workload c4 --batch-size 50
  run c4_strong_b100 --workload c4
done
python - "$OUT" <<'PY'
import glob, json, os, sys
rows = {}
for p in sorted(glob.glob(os.path.join(sys.argv[1], '*_n*.json'))):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:
        continue
    name, n = os.path.basename(p)[:-5].rsplit('_n', 1)
    rows.setdefault(name, {})[int(n)] = d['value']
with open(os.path.join(sys.argv[1], 'table.txt'), 'w') as f:
    for name, r in rows.items():
        base = r.get(1)
        f.write(name + ': ' + '  '.join(f'N={n}: {v:.2f}' + (f' ({v / (base * n):.2f} of linear)' if base and name.startswith("c2") else (f' ({v / base:.2f}x)' if base else ''))
                                       for n, v in sorted(r.items())) + '\n')
print(open(os.path.join(sys.argv[1], 'table.txt')).read())
PY
