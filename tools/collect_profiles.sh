#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: per-kernel time and HBM traffic of the bench command.
#   pass 1: rocprofv3 --kernel-trace --stats          -> gpurun_out/prof/stats (and stats_serial with FASTVLA_OVERLAP=0)
#   pass 2: rocprofv3 --pmc FETCH_SIZE --kernel-trace -> gpurun_out/prof/fetch   (counter passes on their own, as the
#   pass 3: rocprofv3 --pmc WRITE_SIZE --kernel-trace -> gpurun_out/prof/write    pool requires; TCC has no room for both)
# tools/pmc_traffic.py then folds them into profiles/<tag>_kernel_stats.csv and profiles/pmc_traffic.json.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-train --no-train-unfrozen --no-cpu-baseline --no-surface --no-alt > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
echo "stats pass done"
# the same command with the decoder on the tower's stream (FASTVLA_OVERLAP=0): per-kernel durations then add up to the step
# and are the ones bench.py's own hipEvent profile pass (overlap off) must agree with; with overlap on, kernels of the two
# streams share the chip and each one's duration includes the time it waits for CUs
export FASTVLA_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_serial" -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-train --no-train-unfrozen --no-cpu-baseline --no-surface --no-alt > "$OUT/bench_under_rocprof_serial.json" 2> "$OUT/stats_serial.err"
unset FASTVLA_OVERLAP
echo "serial stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --profile-steps 1 --no-train --no-train-unfrozen --no-cpu-baseline --no-surface --no-alt > /dev/null 2> "$OUT/fetch.err"
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --profile-steps 1 --no-train --no-train-unfrozen --no-cpu-baseline --no-surface --no-alt > /dev/null 2> "$OUT/write.err"
echo "write pass done"
# keep what travels back small: the per-dispatch counter tables are tens of MB
python3 "$ROOT/tools/pmc_traffic.py" "$OUT" "$OUT/summary"
rm -rf "$OUT/fetch" "$OUT/write"
find "$OUT/stats" "$OUT/stats_serial" -name '*kernel_trace.csv' -delete
ls -R "$OUT" | head -30
