#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-timeline}
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python tools/c4_per_pocket.py > "$OUT/c4_per_pocket.json" 2> "$OUT/c4.err"; python -c "
import json; d=json.load(open('$OUT/c4_per_pocket.json')); print(d['total_ms_per_step'], d['one_gpu_ligands_per_s'], d['predicted_max_over_mean'])"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tl" -o t -- python "$ROOT/bench.py" --no-cpu-baseline --no-full-run --no-stateless --steps 5 > /dev/null 2> "$OUT/tl.log"
python "$ROOT/tools/step_timeline.py" "$OUT/tl" > "$OUT/timeline_c2.txt"; rm -rf "$OUT/tl"; head -60 "$OUT/timeline_c2.txt"
