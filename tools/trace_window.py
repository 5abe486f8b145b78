"""Timeline window of a rocprofv3 kernel_trace.csv of bench.py, all queues side by side.

python tools/trace_window.py <kernel_trace.csv> [calls_from_end=6] [calls_shown=2]
Prints every fmr kernel from the start of the chosen front-end kernel over `calls_shown` front-end periods: start (us,
relative), duration, queue, name.  Also the front-end start-to-start periods of the whole run.
"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "fmr::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 6
shown = int(sys.argv[3]) if len(sys.argv) > 3 else 2
fe = [i for i, r in enumerate(rows) if "k_ifr_fused" in r["Kernel_Name"] or "k_ifr_decim" in r["Kernel_Name"]]
starts = [int(rows[i]["Start_Timestamp"]) for i in fe]
per = [(b - a) / 1e3 for a, b in zip(starts, starts[1:])]
print("front-end periods us (last 40):", " ".join("%.0f" % p for p in per[-40:]))
print("front-end durations us (last 40):", " ".join("%.0f" % ((int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3) for i in fe[-40:]))
i0 = fe[-back - 1]
t0 = int(rows[i0]["Start_Timestamp"])
t1 = int(rows[fe[-back - 1 + shown]]["Start_Timestamp"]) if back + 1 - shown > 0 else int(rows[-1]["End_Timestamp"])
qs = sorted({r["Queue_Id"] for r in rows})
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e < t0 or s > t1:
        continue
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmr::", "")
    col = qs.index(r["Queue_Id"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{r['Queue_Id']:>3s} " + "    " * col + n[:48])
