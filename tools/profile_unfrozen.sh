#!/bin/bash
# Run on the GPU box from the repo root: per-kernel time of the unfrozen decoder step (tower frozen), B = 32, 320 tokens.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_unfrozen
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/tools/train_unfrozen_bench.py" --steps 5 --warmup 2 > "$OUT/bench.json" 2> "$OUT/stats.err"
find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/stats" -name '*kernel_trace.csv' -exec cp {} "$OUT/kernel_trace.csv" \;
rm -rf "$OUT/stats"
cat "$OUT/bench.json"
