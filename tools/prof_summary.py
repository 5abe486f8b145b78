"""Summarise a rocprofv3 kernel_stats.csv (our kernels only): python tools/prof_summary.py <csv> [steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in rows:
    n = r["Name"]
    if "fmr::" not in n:
        continue
    short = n.split("(")[0].replace("void ", "")
    t = float(r["TotalDurationNs"]) / 1e3
    tot += t
    print(f"{short:36s} calls={int(r['Calls']):5d} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:8.1f} "
          f"max_us={float(r['MaxNs'])/1e3:9.1f} per_step_us={t/steps:9.1f}")
print(f"total per step: {tot/steps:.1f} us")
