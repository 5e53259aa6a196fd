#!/bin/bash
# per-kernel SQ counters of the unfrozen training step (one counter group per pass, counters only: no other trace domains)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_attn
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/a" -- python3 "$R/tools/train_unfrozen_bench.py" --steps 1 --warmup 1 --no-tower > "$OUT/a.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + "/a/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
with open(out + "/summary.txt", "w") as o:
    for k in agg:
        if "attn" in k or "attention" in k:
            o.write(k + f"  launches {n[k]}\n")
            for c, v in sorted(agg[k].items()): o.write(f"    {c:28s} {v / max(n[k], 1):16.0f} per launch\n")
print(open(out + "/summary.txt").read())
PY
rm -rf "$OUT/a"
