"""Copy the summaries tools/collect_profiles.sh left under gpurun_out/<tag>_* into profiles/ and write
profiles/<tag>_summary.json (the numbers DESIGN.md section 7 quotes).   python tools/finish_profiles.py r02"""
import csv, glob, json, os, shutil, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = lambda *a: os.path.join(root, "gpurun_out", *a)
prof = lambda *a: os.path.join(root, "profiles", *a)
os.makedirs(prof(), exist_ok=True)


def bench_line(path):
    if not os.path.exists(path):
        return None
    out = None
    for line in open(path):
        if line.startswith("{"):
            out = json.loads(line)
    return out


bench = bench_line(g(f"{tag}_bench.json"))
json.dump(bench, open(prof(f"{tag}_bench.json"), "w"), indent=1)
stats = g(f"{tag}_stats", f"{tag}_kernel_stats.csv")
shutil.copy(stats, prof(f"{tag}_kernel_stats.csv"))
blocks = str(bench["config"]["blocks_per_step"])
summ = os.path.join(root, "tools", "summarize_pmc.py")
subprocess.check_call([sys.executable, summ, g(f"{tag}_pmc_fetch", "f_counter_collection.csv"),
                       g(f"{tag}_pmc_write", "w_counter_collection.csv"), prof(f"{tag}_pmc_traffic.json"), blocks])
if os.path.exists(g(f"{tag}_pmc_fetch_nofused", "f_counter_collection.csv")):
    subprocess.check_call([sys.executable, summ, g(f"{tag}_pmc_fetch_nofused", "f_counter_collection.csv"),
                           g(f"{tag}_pmc_write_nofused", "w_counter_collection.csv"),
                           prof(f"{tag}_pmc_traffic_three_kernel_front_end.json"), blocks])
for extra in (f"{tag}_pytest_gpu.log", f"{tag}_smoke.log"):
    if os.path.exists(g(extra)):
        shutil.copy(g(extra), prof(extra))
for rep in ("parity_report.json", "parity_report_configs.json"):
    if os.path.exists(g(rep)):
        shutil.copy(g(rep), prof(f"{tag}_{rep}"))
others = {}
for p in sorted(glob.glob(g(f"{tag}_bench_*.json"))):
    b = bench_line(p)
    if b is None:
        continue
    name = os.path.basename(p)
    json.dump(b, open(prof(name), "w"), indent=1)
    others[name[len(tag) + 7:-5]] = {"value": b["value"], "unit": b["unit"], "ms_per_step": b["ms_per_step"],
                                     "stage_ms": b["roofline"].get("stage", {}).get("ms") if isinstance(b["roofline"].get("stage"), dict) else None}

# the dominant kernel in the rocprofv3 --stats run of the same command
rows = list(csv.DictReader(open(stats)))
fused = [r for r in rows if "k_ifr_fused" in r["Name"]]
dom = fused[0] if fused else [r for r in rows if "k_ifr_decim" in r["Name"]][0]
trace = g(f"{tag}_stats", f"{tag}_kernel_trace.csv")
key = "k_ifr_fused" if fused else "k_ifr_decim"
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(trace)) if key in r["Kernel_Name"]]
med = sorted(durs)[len(durs) // 2]
full = [d for d in durs if d > 0.97 * med]                     # bench.py's set-up call is cut in two shorter launches
ev = bench_line(g(f"{tag}_stats.log"))                          # the profiled run's own bench line (HIP events in the timed region)
summary = {
    "bench": {k: bench[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup")},
    "roofline": bench["roofline"], "cpu_baseline": bench.get("cpu_baseline"),
    "dominant_kernel": key,
    "rocprofv3_stats": {"calls": int(dom["Calls"]), "average_us_all_launches": float(dom["AverageNs"]) / 1e3,
                        "average_us_full_batch_launches": sum(full) / len(full), "full_batch_launches": len(full)},
    "hip_events_profiled_run_ms": ev["roofline"].get("avg_launch_ms") if ev else None,
    "pmc_traffic": json.load(open(prof(f"{tag}_pmc_traffic.json"))),
    "other_configs": others,
}
json.dump(summary, open(prof(f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:6000])
