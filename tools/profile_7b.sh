#!/bin/bash
# Run on the GPU box from the repo root: per-kernel time of the 7B, B = 16 step (C4), streams serialised.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_7b
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export FASTVLA_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --model fastvlm-7b --batch 16 --llm-precision 1 --steps 6 --warmup 2 --no-train --no-train-unfrozen --no-cpu-baseline --no-surface --no-alt > "$OUT/bench.json" 2> "$OUT/err.txt"
find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/stats"
cut -c1-200 "$OUT/bench.json"
