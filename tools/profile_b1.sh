set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_b1
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export FASTVLA_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --batch ${B1:-1} --steps 20 --warmup 3 --no-train --no-train-unfrozen --no-cpu-baseline --no-surface --no-alt > "$OUT/bench.json" 2> "$OUT/err.txt"
find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/stats" -name '*kernel_trace.csv' -exec cp {} "$OUT/kernel_trace.csv" \;
rm -rf "$OUT/stats"
unset FASTVLA_OVERLAP
python3 "$ROOT/tools/latency_small_batch.py" > "$OUT/latency.txt" 2>&1
cat "$OUT/latency.txt"
cut -c1-300 "$OUT/bench.json"
