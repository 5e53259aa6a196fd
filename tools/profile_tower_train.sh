#!/bin/bash
# Run on the GPU box from the repo root: per-kernel time of the tower's training forward + backward alone (B = 32, fixed dL/d tower_out).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_tower
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/tools/train_unfrozen_bench.py" --train-tower --tower-only --steps 6 --warmup 2 > "$OUT/bench.json" 2> "$OUT/stats.err"
find "$OUT/stats" -name '*kernel_trace.csv' -delete
find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
cat "$OUT/bench.json"
