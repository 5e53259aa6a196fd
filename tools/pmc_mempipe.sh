#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: memory-pipeline counters per kernel of the bench command, one
# rocprofv3 --pmc pass per counter group (counter passes carry --kernel-trace only, as the pool requires).
#   gpurun_out/mempipe/<group>.csv : kernel, launches, average of each counter per launch
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/mempipe
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp FASTVLA_OVERLAP=0
i=0
# one counter per pass: a block's counter slots are few (two TA counters together already "exceed the capabilities of the
# hardware"), and a refused configuration leaves rocprofv3 hanging after its abort -- hence the timeout on every pass
for group in "TA_TA_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum" "TCC_MISS" "TCC_HIT"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/g$i" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --profile-steps 1 --no-train --no-cpu-baseline > /dev/null 2> "$OUT/g$i.err" || echo "group $i failed"
  python3 - "$OUT/g$i" "$OUT/g$i.csv" <<'PY'
import csv, glob, sys, collections
src, dst = sys.argv[1], sys.argv[2]
files = glob.glob(src + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("fv::(anonymous namespace)::", "").split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
names = sorted({c for k in acc for c in acc[k]})
with open(dst, "w") as o:
    o.write("kernel,launches," + ",".join(names) + "\n")
    for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
        o.write(k + "," + str(len(n[k])) + "," + ",".join("%.0f" % (acc[k][c] / max(len(n[k]), 1)) for c in names) + "\n")
PY
  rm -rf "$OUT/g$i"
  echo "group $i done"
done
