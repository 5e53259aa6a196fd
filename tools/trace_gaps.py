"""tools/trace_gaps.py <kernel_trace.csv> -- where the GPU is idle inside the timed steps of a `rocprofv3 --kernel-trace` run of bench.py:
union of the kernels' busy intervals against the wall span, per step (steps are cut at the letterbox kernel), and the gaps by the
kernel they follow.  Run on the GPU box; prints a short table."""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r"(?:::|^|\s)(\w+)(<[^(]*>)?\((?!anonymous)", name)
    return (m.group(1) + (m.group(2) or ""))[:60] if m else name[:60]

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("letterbox_kernel")]
if len(starts) < 3:
    print("no step markers", len(rows))
    sys.exit(0)
steps = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
print(f"{len(rows)} dispatches, {len(steps)} whole steps")
for si, (a, b) in enumerate(steps[-4:]):
    seg = rows[a:b]
    t0, t1 = seg[0][0], rows[b][0]
    busy, cur_e, gaps = 0, t0, collections.Counter()
    last = None
    for s, e, n, q in seg:
        if s > cur_e:
            gaps[last] += s - cur_e
            busy += 0
        if e > cur_e:
            busy += e - max(s, cur_e)
            cur_e = e
            last = n
    wall = t1 - t0
    print(f"step {si}: wall {wall/1e6:.2f} ms, busy union {busy/1e6:.2f} ms, idle {100*(wall-busy)/wall:.1f} %, kernels {len(seg)}, sum of durations {sum(e-s for s,e,_,_ in seg)/1e6:.2f} ms")
    if si == len(steps[-4:]) - 1:
        for n, g in gaps.most_common(12):
            print(f"   idle after {n:70s} {g/1e3:8.1f} us")
