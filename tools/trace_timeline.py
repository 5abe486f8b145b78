"""Timeline of one steady-state chain call from a rocprofv3 kernel_trace.csv.

python tools/trace_timeline.py <kernel_trace.csv> [call_index_from_end=3]
Prints every fmr kernel between the chosen call's front-end kernel and the next call's, with
start (us, relative), duration and queue id, plus the decim durations of all calls.
"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "fmr::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dec = [i for i, r in enumerate(rows) if "k_ifr_decim" in r["Kernel_Name"] or "k_ifr_fused" in r["Kernel_Name"]]
print("decim us:", " ".join("%.0f" % ((int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3) for i in dec))
starts = [int(rows[i]["Start_Timestamp"]) for i in dec]
print("decim start-to-start us:", " ".join("%.0f" % ((b - a) / 1e3) for a, b in zip(starts, starts[1:])))
i0, i1 = dec[-back - 1], dec[-back]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmr::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{r['Queue_Id']:>3s}  {n[:60]}")
