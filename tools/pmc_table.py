"""Per-kernel averages of every counter in rocprofv3 counter_collection CSVs.
python tools/pmc_table.py <csv> [<csv> ...] [--filter substr]"""
import csv, sys
flt = None
files = []
a = sys.argv[1:]
while a:
    x = a.pop(0)
    if x == "--filter":
        flt = a.pop(0)
    else:
        files.append(x)
agg = {}
for fn in files:
    for r in csv.DictReader(open(fn)):
        n = r["Kernel_Name"]
        if "fmr" not in n or (flt and flt not in n):
            continue
        n = n.split("(")[0].replace("void ", "").replace("fmr::", "")
        d = agg.setdefault(n, {}).setdefault(r["Counter_Name"], [0, 0.0])
        d[0] += 1
        d[1] += float(r["Counter_Value"])
for n in sorted(agg):
    print(n)
    for c, (k, v) in sorted(agg[n].items()):
        print(f"    {c:28s} {v / k:16.1f}   (n={k})")
