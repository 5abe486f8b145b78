"""Copy the summaries tools/collect_profiles.sh left under gpurun_out/<tag>_* into profiles/, write
profiles/<tag>_summary.json and refresh the measured rows of DESIGN.md section 7.   python tools/finish_profiles.py r02"""
import csv, glob, json, os, re, shutil, subprocess, sys
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
# the bench line of the collection ran before this summary existed: its traffic field is filled from the same collection
_pm = json.load(open(prof(f"{tag}_pmc_traffic.json")))
_fk = [v for k, v in _pm["kernels"].items() if "k_ifr_fused" in k]
if _fk and bench["roofline"].get("traffic") is None:
    bench["roofline"]["traffic"] = _fk[0]["hbm_bytes"]
    bench["roofline"]["traffic_source"] = (f"profiles/{tag}_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same "
                                           f"collection (kernel sources {_pm.get('csrc_sha256_16')}), filled in by tools/finish_profiles.py")
    json.dump(bench, open(prof(f"{tag}_bench.json"), "w"), indent=1)
for extra in (f"{tag}_pytest_gpu.log", f"{tag}_smoke.log", f"{tag}_fused_harness.log", f"{tag}_fe_stamps.txt", f"{tag}_am_agc_rounds.txt",
              f"{tag}_block1_trace.txt", f"{tag}_step_time.txt", f"{tag}_pmc_r8b_stage_b.txt", f"{tag}_step_timeline.txt",
              f"{tag}_step_timeline_r8b.txt", f"{tag}_step_timeline_if_filter.txt", f"{tag}_step_timeline_sigma_1e-2.txt",
              f"{tag}_mpf_account.txt", f"{tag}_pll_mismatch.txt", f"{tag}_step_timeline_am.txt", f"{tag}_kernel_stats_r8b.csv",
              f"{tag}_kernel_stats_if_filter.csv", f"{tag}_kernel_stats_am.csv"):
    if os.path.exists(g(extra)):
        shutil.copy(g(extra), prof(extra))
if os.path.exists(g(f"{tag}_pmc_fetch_r8b", "f_counter_collection.csv")):
    subprocess.check_call([sys.executable, summ, g(f"{tag}_pmc_fetch_r8b", "f_counter_collection.csv"),
                           g(f"{tag}_pmc_write_r8b", "w_counter_collection.csv"), prof(f"{tag}_pmc_traffic_r8b.json"), blocks])
for rep in ("parity_report.json", "parity_report_configs.json", "parity_report_levels.json"):
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
                                     "stage_ms": (b.get("roofline") or {}).get("stage", {}).get("ms") if isinstance((b.get("roofline") or {}).get("stage"), dict) else None}

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

# ---- rows keyed by their first column
rf, cb = bench["roofline"], bench.get("cpu_baseline") or {}
pm = summary["pmc_traffic"]["kernels"]
fk = [v for k, v in pm.items() if "k_ifr_fused" in k]
alg = rf["algorithmic_bytes_per_launch"]
o = lambda name: others.get(name) or {}
nf = bench_line(g(f"{tag}_bench_nofused.json"))
pm3 = json.load(open(prof(f"{tag}_pmc_traffic_three_kernel_front_end.json")))["kernels"] if os.path.exists(prof(f"{tag}_pmc_traffic_three_kernel_front_end.json")) else {}
three = sum(v["hbm_bytes"] for k, v in pm3.items() if any(n in k for n in ("k_ifr_decim", "k_ifr_poly", "k_disc"))
            and "k_ifr_poly5h" not in k and "k_ifr_decim2<128, 24" not in k)      # (the run's r8b leg has its own stage A / B kernels)
gs = lambda v: f"{v / 1e3:.1f} GS/s" if v and v >= 1e4 else (f"{v / 1e3:.2f} GS/s" if v and v >= 1e3 else f"{v:.1f} MS/s")
rs = summary["rocprofv3_stats"]
rows = {
    "whole-job throughput, config 2 (1 stream)": f"{gs(bench['value'])} ({bench['ms_per_step']:.3f} ms per 2^27-sample step)",
    "`k_ifr_fused` average launch (HIP events in the timed region / rocprofv3)":
        f"{rf['avg_launch_ms'] * 1e3:.1f} µs (bench run) / {summary['hip_events_profiled_run_ms'] * 1e3:.1f} µs vs {rs['average_us_full_batch_launches']:.1f} µs (events vs rocprofv3 kernel trace, the {rs['full_batch_launches']} full-batch launches of the profiled run; `--stats` average over all {rs['calls']} launches incl. the shorter set-up launches: {rs['average_us_all_launches']:.1f} µs)",
    "`roofline` (HBM, 8 B × 2^27 per launch ÷ launch time ÷ 8 TB/s) — dominant kernel = the whole FIR + discriminator stage": f"{rf['achieved']:.0f} GB/s = **{rf['frac']:.3f}** of peak",
    "same stage with the three-kernel front end (`FMR_NO_FUSED=1`, round 1's path, same box)":
        (f"{nf['roofline']['stage']['ms']:.3f} ms = {nf['roofline']['stage']['frac']:.3f}; whole job {gs(nf['value'])}" if nf else "not collected"),
    "PMC traffic of the stage (FETCH_SIZE×2 + WRITE_SIZE, separate passes)":
        (f"{fk[0]['hbm_bytes'] / 1e9:.3f} GB per launch = {fk[0]['hbm_bytes'] / alg:.3f} × algorithmic ({fk[0]['read_bytes'] / 1e9:.3f} GB read, {fk[0]['write_bytes'] / 1e9:.3f} GB write: MPX + |x|^2 as floats)" if fk else "n/a")
        + (f"; three-kernel front end {three / 1e9:.3f} GB = {three / alg:.2f} ×" if three else ""),
    "CPU oracle (`cpu_baseline`, kind \"port\"), same stream":
        (f"{cb.get('value', 0):.1f} MS/s on 1 core; {((cb.get('all_cores') or {}).get('value') or 0):.0f} MS/s with one oracle process per core ({(cb.get('all_cores') or {}).get('cores', '?')} cores)" if cb else "n/a"),
    "audio check inside the bench run (stream 0, first call, vs oracle)": f"RMS error {bench['audio_check'].get('audio_rms_err_vs_oracle')} over {bench['audio_check'].get('audio_samples_checked')} samples (tolerance 1e-5)",
    "config 5 shard: 32 FM stereo streams per GPU": (f"{gs(o('config5_32streams').get('value'))} ({o('config5_32streams').get('ms_per_step')} ms per step of 32 × 128 blocks)" if o('config5_32streams') else "n/a"),
    "config 4: `-E 64`, one stream / 32 streams": (f"{gs(o('config4_E64').get('value'))} / {gs(o('config4_E64_32streams').get('value'))}"
        + (f"; 128 / 256 streams (one equaliser workgroup per stream: the 256 CUs fill up) {gs(o('config4_E64_128streams').get('value'))} / {gs(o('config4_E64_256streams').get('value'))}" if o('config4_E64_256streams') else "")
        if o('config4_E64') else "n/a"),
    "config 3: AM 384 kS/s → 48 k, one stream / 32 streams": (f"{gs(o('config3_am').get('value'))} / {gs(o('config3_am_32streams').get('value'))}" if o('config3_am') else "n/a"),
    "stereo decoder on a mono station (unlocked PLL, serial fallback)": ((f"{gs(o('no_pilot').get('value'))}"
        + (f" one stream; 64 / 256 streams per GPU (the serial loop runs one stream per lane: 64 streams cost one wave what one costs) {gs(o('no_pilot_64streams').get('value'))} / {gs(o('no_pilot_256streams').get('value'))}" if o('no_pilot_256streams') else ""))
        if o('no_pilot') else "n/a"),
}
pd = os.path.join(root, "DESIGN.md")
sd = open(pd).read()
for k, v in rows.items():
    sd, n = re.subn(r"^\| " + re.escape(k) + r" \|.*\|$", lambda m: f"| {k} | {v} |", sd, flags=re.M)
    assert n == 1, k
open(pd, "w").write(sd)
