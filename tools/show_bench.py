import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["roofline"]["avg_launch_ms"], d["recurrences"], {k:v for k,v in d["kernel_ms_per_step"].items() if k in ("pll","pilotcut","pll_begin")})
    elif "host prof" in l: print(l.strip())
