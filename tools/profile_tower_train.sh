set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_tower
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/tools/train_unfrozen_bench.py" --train-tower --tower-only --steps 3 --warmup 1 > "$OUT/bench.json" 2> "$OUT/stats.err"
find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
python3 - "$OUT" <<'PY'
import csv,glob,sys,re,os
out=sys.argv[1]
f=glob.glob(os.path.join(out,'stats','**','*kernel_trace.csv'),recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
with open(os.path.join(out,'trace_compact.csv'),'w') as g:
    for r in rows:
        n=re.sub(r'\(.*','',r['Kernel_Name'].replace('fv::(anonymous namespace)::','').replace('void ',''))
        g.write(f"{n},{int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X']))},{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:.1f}\n")
PY
rm -rf "$OUT/stats"
cat "$OUT/bench.json"
