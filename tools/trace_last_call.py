"""Per-kernel totals of the LAST chain call in a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "fmr::" in r["Kernel_Name"]]
idx = [i for i, r in enumerate(rows) if "k_ifr_decim" in r["Kernel_Name"]][-1]
last = rows[idx:]
t0, tend = int(last[0]["Start_Timestamp"]), int(last[-1]["End_Timestamp"])
agg = {}
for r in last:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for n, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"{n:36s} calls={c:3d} total_us={d:8.1f}")
print("sum kernels us %.1f  wall us %.1f  launches %d" % (sum(d for c, d in agg.values()), (tend - t0) / 1e3, len(last)))
