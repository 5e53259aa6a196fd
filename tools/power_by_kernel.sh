#!/bin/bash
# on the GPU box: board power / sclk per kernel family (tools/power_by_kernel.py) -> gpurun_out/power_by_kernel.txt
cd "$(dirname "$0")/.."
M=gpurun_out/pbk_marks.log; S=gpurun_out/pbk_samples.log; : > $M; : > $S
# optional arguments: family list (comma separated), then variant libraries "name:path" whose MARKs carry the name as a suffix
FAM=$1; shift
(
  if [ -z "$FAM" ]; then timeout -k 5 200 python tools/power_by_kernel.py 4 2>&1 | grep MARK >> $M
  else for v in "$@"; do FASTVLA_HIP_LIB=${v#*:} timeout -k 5 100 python tools/power_by_kernel.py 3 $FAM _${v%%:*} 2>&1 | grep MARK >> $M; done; fi
) &
BP=$!
while kill -0 $BP 2>/dev/null; do
  t=$(date +%s.%N)
  r=$(rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //; s/[()A-Za-z ]//g' | tr '\n' ' ')
  echo "$t $r" >> $S
done
wait $BP
python3 - <<'PY' > gpurun_out/power_by_kernel.txt
import statistics
marks = [l.split() for l in open("gpurun_out/pbk_marks.log")]
samples = []
for l in open("gpurun_out/pbk_samples.log"):
    p = l.split()
    if len(p) >= 3:
        try: samples.append((float(p[0]), float(p[1].strip(":")), float(p[2])))
        except ValueError: pass
for m in marks:
    name, t0, t1 = m[1], float(m[2]), float(m[3])
    inside = [(a, b) for t, a, b in samples if t0 + 0.7 < t < t1 - 0.3]
    if inside:
        print(f"{name:28s} {' '.join(m[4:]):36s} sclk {statistics.median(a for a, b in inside):6.0f} MHz  power {statistics.median(b for a, b in inside):6.0f} W  ({len(inside)} samples)")
    else:
        print(f"{name:28s} {' '.join(m[4:]):36s} no samples")
PY
cat gpurun_out/power_by_kernel.txt
