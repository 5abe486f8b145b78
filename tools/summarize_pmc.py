"""Reduce the rocprofv3 --pmc counter_collection CSVs (separate FETCH_SIZE / WRITE_SIZE passes) to a
small per-kernel JSON under profiles/.  gfx950 correction (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read, i.e. HALF the bytes -> doubled;
WRITE_SIZE is used as reported.  Units of both counters: KiB."""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def per_kernel(fn):
    agg = {}
    for r in csv.DictReader(open(fn)):
        if "fmr::" not in r["Kernel_Name"]:
            continue
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg.setdefault(n, []).append(float(r["Counter_Value"]))
    # median over launches: the bench's set-up call is cut in two shorter launches (cold start), every other
    # launch has the full batch
    return {k: sorted(v)[len(v) // 2] for k, v in agg.items()}

fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline",
       "blocks_per_step": blocks, "csrc_sha256_16": __import__("bench").csrc_hash(), "units": "bytes per launch (median over launches)",
       "correction": "read bytes = 2 * FETCH_SIZE * 1024 (gfx950 wide-read under-count), write bytes = WRITE_SIZE * 1024",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    rd, wr = 2 * fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
    out["kernels"][k] = {"fetch_size_kib_raw": fetch.get(k, 0.0), "write_size_kib_raw": write.get(k, 0.0),
                         "read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k in out["kernels"]:
    if "k_ifr_fused" in k or "k_ifr_decim" in k or "k_ifr_poly" in k or "k_disc" in k:
        print(k, out["kernels"][k])
