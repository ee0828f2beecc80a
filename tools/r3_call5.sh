#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03e
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python tools/wg_balance.py > "$OUT/wg_balance_c2.txt" 2> "$OUT/wg_balance_c2.err"; cat "$OUT/wg_balance_c2.txt"; tail -3 "$OUT/wg_balance_c2.err"
timeout 900 python -m pytest tests/test_gpu_long_parity.py -m gpu -q -x -k "launcher or rebuild" > "$OUT/pytest_gpu.txt" 2>&1; tail -5 "$OUT/pytest_gpu.txt"
