import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["roofline"]["avg_launch_ms"], d["recurrences"], {k:v for k,v in d["kernel_ms_per_step"].items() if k not in ("ifr_decim","ifr_poly","if_agc","stats","pll_finish")})
    elif "host prof" in l: print(l.strip())
