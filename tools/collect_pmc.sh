#!/bin/bash
# the two counter passes of tools/collect_profiles.sh on their own -> gpurun_out/prof/summary/pmc_traffic.json
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT/fetch" "$OUT/write"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --profile-steps 1 --no-train --no-cpu-baseline --no-surface --no-alt > /dev/null 2> "$OUT/fetch.err"
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --profile-steps 1 --no-train --no-cpu-baseline --no-surface --no-alt > /dev/null 2> "$OUT/write.err"
echo "write pass done"
python3 "$ROOT/tools/pmc_traffic.py" "$OUT" "$OUT/summary" > /dev/null
rm -rf "$OUT/fetch" "$OUT/write"
