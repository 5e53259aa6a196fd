"""rocprofv3 --kernel-trace CSV -> one line per launch of the LAST optimiser step (duration, grid, kernel): which GEMM shapes are slow."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
lo = idx[-2] + 1 if len(idx) > 1 else 0
hi = idx[-1] if idx else len(rows) - 1
gx = "Grid_Size_X" if "Grid_Size_X" in rows[0] else ("Grid_Size" if "Grid_Size" in rows[0] else None)
with open(sys.argv[2], "w") as out:
    out.write("# columns: " + ",".join(rows[0].keys()) + "\n")
    for r in rows[lo:hi + 1]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        out.write(f"{d:9.1f} us grid {r.get(gx, '?'):>8}  {r['Kernel_Name'][:100]}\n")
