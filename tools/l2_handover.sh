#!/bin/bash
# Run on the GPU box from the repo root: tools/l2_handover.hip timed, then under the two PMC passes (per-dispatch FETCH_SIZE / WRITE_SIZE).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/l2_handover
rm -rf "$OUT"; mkdir -p "$OUT"
BIN=$ROOT/tools/bin/l2_handover
[ -x "$BIN" ] || hipcc -O3 --offload-arch=gfx950 "$ROOT/tools/l2_handover.hip" -o "$BIN"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 "$BIN" 32 64 > "$OUT/timing.txt"
cat "$OUT/timing.txt"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- "$BIN" 32 64 > /dev/null 2> "$OUT/fetch.err"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- "$BIN" 32 64 > /dev/null 2> "$OUT/write.err"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
def rows(sub, name):
    acc = {}
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            k = int(r["Dispatch_Id"])
            kn = r["Kernel_Name"].replace("void ", "")
            acc.setdefault(k, [kn[:40], 0.0])[1] += float(r["Counter_Value"])
    return acc
fe, wr = rows("fetch", "FETCH_SIZE"), rows("write", "WRITE_SIZE")
with open(os.path.join(out, "pmc.txt"), "w") as f:
    for k in sorted(set(fe) | set(wr)):
        name = (fe.get(k) or wr.get(k))[0]
        line = f"dispatch {k:3d} {name:40s} fetch {2 * fe.get(k, [0, 0])[1] / 1024:9.1f} MB (2 x FETCH_SIZE)   write {wr.get(k, [0, 0])[1] / 1024:9.1f} MB"
        print(line); f.write(line + "\n")
PY
rm -rf "$OUT/fetch" "$OUT/write"
