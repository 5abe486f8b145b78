"""Where and when the workgroups of the PLL integration passes ran (diagnostic build -DFMR_PLL_TRACE).

python tools/pll_trace.py <dump.bin>
Records: per workgroup and pass (0 = Jacobian pass, 1 = later pass) {100 MHz wall clock at start, shader cycles spent,
XCC_ID<<32 | HW_ID, 100 MHz wall clock at end}.  HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13.
"""
import sys, collections
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64)
a = a[: len(a) // 8 * 8].reshape(-1, 2, 4)
for p, name in ((0, "Jacobian pass"), (1, "later pass")):
    r = a[:, p, :]
    ok = r[:, 3] != 0
    r = r[ok]; wg = np.nonzero(ok)[0]
    if not len(r):
        continue
    hw = (r[:, 2] & 0xFFFFFFFF).astype(np.int64); xcc = (r[:, 2] >> 32).astype(np.int64) & 0xF
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    t0 = r[:, 0].astype(np.float64); t1 = r[:, 3].astype(np.float64); base = t0.min()
    t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0     # us
    cyc = r[:, 1].astype(np.float64)
    print(f"{name}: {len(r)} workgroups")
    print(f"   start (us after the first start): median {np.median(t0):.1f} p90 {np.percentile(t0,90):.1f} max {t0.max():.1f}")
    print(f"   lifetime us: min {np.min(t1-t0):.1f} median {np.median(t1-t0):.1f} max {np.max(t1-t0):.1f};  shader cycles median {np.median(cyc):.0f} -> {np.median(cyc/(t1-t0)):.0f} MHz")
    print(f"   end: median {np.median(t1):.1f} max {t1.max():.1f}")
    o = np.argsort(wg)
    print("   start by workgroup index (every 128th):", " ".join(f"{t0[o][i]:.1f}" for i in range(0, len(o), 128)))
    key = list(zip(xcc, se, sh, cu, simd))
    per_simd = collections.Counter(key)
    load = np.array([per_simd[k] for k in key])
    print("   SIMDs used", len(per_simd), "waves per used SIMD:", dict(sorted(collections.Counter(per_simd.values()).items())))
    for l in sorted(set(load.tolist())):
        d = (t1 - t0)[load == l]
        print(f"   SIMD load {l}: {len(d)} waves, lifetime median {np.median(d):.1f} us max {d.max():.1f}")
