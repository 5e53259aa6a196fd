#!/bin/bash
# VERDICT r5 #1's table, on the GPU box from the repo root (build first, here: tools/dw_variants.sh nodw7:"-DDP_NO_T7" && hipcc ... tools/dw7_prologue.hip):
#   today:     dwpair_march_kernel<64,16> (x -> x', t)            + convffn32_kernel<384> (t, x' -> out)
#   fused:     dwpair without its 7x7 half (x -> x')              + convffn32_kernel<384> + the dw7x7 prologue (x' -> its own B fragments)
# each piece timed on its own at the C = 384 shape of the headline step (B = 64, 64 x 64 x 384), then under the two PMC passes (per-dispatch
# FETCH_SIZE / WRITE_SIZE; FETCH_SIZE doubled per MI355X_MICROARCH.md).  -> gpurun_out/dw7_prologue/table.txt
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/dw7_prologue
rm -rf "$OUT"; mkdir -p "$OUT"
BIN=$ROOT/tools/bin/dw7_prologue
[ -x "$BIN" ] || hipcc -O3 -std=c++17 --offload-arch=gfx950 "$ROOT/tools/dw7_prologue.hip" -o "$BIN"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 "$BIN" 64 20 > "$OUT/prologue_timing.txt"
cat "$OUT/prologue_timing.txt"
# the library's kernels at the same shape, same box: the pair as shipped, the pair without its 7x7 half, the fused ConvFFN
( cd "$ROOT" && timeout -k 10 300 python3 tools/dwpair_bench.py 4 > "$OUT/dwpair_today.txt" 2>&1 ) || { cat "$OUT/dwpair_today.txt"; exit 1; }
( cd "$ROOT" && FASTVLA_HIP_LIB=tools/bin/libdw_nodw7.so timeout -k 10 300 python3 tools/dwpair_bench.py 4 > "$OUT/dwpair_nodw7.txt" 2>&1 ) || { cat "$OUT/dwpair_nodw7.txt"; exit 1; }
( cd "$ROOT" && timeout -k 10 300 python3 tools/ffn_bench.py 4 > "$OUT/ffn_today.txt" 2>&1 ) || { cat "$OUT/ffn_today.txt"; exit 1; }
tail -1 "$OUT/dwpair_today.txt"; tail -1 "$OUT/dwpair_nodw7.txt"; grep "C=384" "$OUT/ffn_today.txt"
# PMC: the prototype (each dispatch on its own row), then the library's pair in both builds
pmc() {   # $1 = tag, rest = command
  local tag=$1; shift
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/$tag.fetch" -- "$@" > /dev/null 2> "$OUT/$tag.fetch.err"
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/$tag.write" -- "$@" > /dev/null 2> "$OUT/$tag.write.err"
}
pmc proto "$BIN" 64 2
cd "$ROOT"
pmc today python3 tools/dwpair_bench.py 1
export FASTVLA_HIP_LIB=tools/bin/libdw_nodw7.so
pmc nodw7 python3 tools/dwpair_bench.py 1
unset FASTVLA_HIP_LIB
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
def rows(tag, sub, name):
    acc = {}
    for f in glob.glob(os.path.join(out, f"{tag}.{sub}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            k = int(r["Dispatch_Id"])
            kn = r["Kernel_Name"].replace("void ", "").replace("fv::(anonymous namespace)::", "")
            acc.setdefault(k, [kn[:46], 0.0])[1] += float(r["Counter_Value"])
    return acc
with open(os.path.join(out, "pmc.txt"), "w") as f:
    for tag in ("proto", "today", "nodw7"):
        fe, wr = rows(tag, "fetch", "FETCH_SIZE"), rows(tag, "write", "WRITE_SIZE")
        seen = {}
        for k in sorted(set(fe) | set(wr)):
            name = (fe.get(k) or wr.get(k))[0]
            if "dw7_prologue" not in name and "dwpair" not in name: continue
            fm, wm = 2 * fe.get(k, [0, 0])[1] / 1024, wr.get(k, [0, 0])[1] / 1024
            key = (name, round(fm, -1), round(wm, -1))
            seen[key] = seen.get(key, 0) + 1
        for (name, fm, wm), n in seen.items():
            line = f"{tag:6s} {name:46s} x{n:3d}  fetch {fm:8.0f} MB (2 x FETCH_SIZE)   write {wm:8.0f} MB"
            print(line); f.write(line + "\n")
PY
rm -rf "$OUT"/*.fetch "$OUT"/*.write
cat "$OUT/prologue_timing.txt" "$OUT/pmc.txt" > "$OUT/table.txt"
( echo "--- dwpair as shipped:"; tail -1 "$OUT/dwpair_today.txt"; echo "--- dwpair without its 7x7 half (-DDP_NO_T7):"; tail -1 "$OUT/dwpair_nodw7.txt"; echo "--- fused ConvFFN as shipped:"; grep "C=" "$OUT/ffn_today.txt" ) >> "$OUT/table.txt"
