#!/usr/bin/env python3
"""Fold the rocprofv3 passes of tools/collect_profiles.sh into the two files the repo keeps under profiles/:
   <out>/kernel_stats.csv  -- the --stats per-kernel table (calls, total / average ns, share)
   <out>/pmc_traffic.json  -- HBM bytes per launch per kernel = (2 * FETCH_SIZE + WRITE_SIZE) KiB * 1024: FETCH_SIZE counts
                              half of wide coalesced read streams on gfx950 (MI355X_MICROARCH.md, HBM section), WRITE_SIZE
                              is exact; averaged over every launch of the kernel.  bench.py reads it for roofline.traffic."""
import csv, glob, json, os, re, sys

def short(name):
    m = re.search(r"(\w+)(<|\()", name.replace("(anonymous namespace)::", "").replace("fv::", "").replace("void ", ""))
    return m.group(1) if m else name[:40]

def instance(name):
    """template kernels also under `name<first template argument>` (convffn32_kernel<384>: one row of the --stats table each)"""
    m = re.search(r"(\w+)<(\d+)", name.replace("(anonymous namespace)::", "").replace("fv::", "").replace("void ", ""))
    return f"{m.group(1)}<{m.group(2)}>" if m else None

def counters(d, want):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != want: continue
            for key in (short(r["Kernel_Name"]), instance(r["Kernel_Name"])):
                if key is None: continue
                a = acc.setdefault(key, {})
                a[r["Dispatch_Id"]] = a.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in acc.items()}

def main(src, out):
    os.makedirs(out, exist_ok=True)
    for sub, name in (("stats", "kernel_stats.csv"), ("stats_serial", "kernel_stats_serial.csv")):
        stats = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
        if stats:
            rows = list(csv.reader(open(stats[0])))
            with open(os.path.join(out, name), "w", newline="") as f:
                csv.writer(f).writerows(rows)
    fe, wr = counters(os.path.join(src, "fetch"), "FETCH_SIZE"), counters(os.path.join(src, "write"), "WRITE_SIZE")
    res = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/collect_profiles.sh) on `bench.py --steps 2 --warmup 1 "
                    "--profile-steps 1 --no-train --no-cpu-baseline`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch (gfx950 FETCH_SIZE "
                    "reads half of wide coalesced streams; MI355X_MICROARCH.md HBM section), averaged over all launches of the kernel",
           "_detail": {}}
    for k in sorted(set(fe) | set(wr)):
        f, nf = fe.get(k, (0.0, 0)); w, nw = wr.get(k, (0.0, 0))
        res[k] = int((2 * f + w) * 1024)
        res["_detail"][k] = {"launches": max(nf, nw), "fetch_kib_raw": round(f, 1), "write_kib": round(w, 1), "hbm_bytes_per_launch": res[k]}
    json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if not k.startswith("_")}, indent=1))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
