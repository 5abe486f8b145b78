#!/usr/bin/env python3
"""What one Newton multiple-shooting round leaves of the IF AGC's state (float64 model, CPU only).

The recurrence is IfSimpleAgc's (sfmbase/IfSimpleAgc.cpp:36-51): g <- g (1 + r (1 - g^2 |x|^2)), r = 1e-4, chunks of 256
samples as in k_agc_round (csrc/kernels_par.hpp).  From a flat start guess (the carried gain everywhere) a round integrates
every chunk with its sensitivity and the node pass solves the linearised boundary conditions; `movement` is what that
pass reports (largest relative change of a node), `error` the largest relative distance of the new nodes from the serial
solution, `error_end` that of the call's end state.  The table this prints is what the round-1 acceptance of the state-only
solve (movement <= 2e-3) rests on: error ~ 1.3 x movement^2.

python tools/agc_round_model.py
"""
import numpy as np

R, C = 1e-4, 256


def rounds(sigma, amp, g0, n=400_000, n_rounds=3, seed=1):
    """[(movement, error, error_end)] of the first n_rounds Newton rounds on amp * exp(j phi) + noise(sigma)."""
    rng = np.random.default_rng(seed)
    x = amp * np.exp(1j * rng.uniform(0, 2 * np.pi, n)) + sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    e = x.real ** 2 + x.imag ** 2
    nc = n // C
    g = g0
    truth = np.empty(nc + 1)
    truth[0] = g
    for c in range(nc):                       # the serial solution
        for i in range(c * C, (c + 1) * C):
            g = g * (1 + R * (1 - g * g * e[i]))
        truth[c + 1] = g
    nodes = np.full(nc + 1, float(g0))
    E = e[:nc * C].reshape(nc, C)
    out = []
    for _ in range(n_rounds):
        g = nodes[:nc].copy()
        dg = np.ones(nc)
        for i in range(C):                    # integration pass, all chunks at once
            nrm = g * g * E[:, i]
            z = 1 + R * (1 - nrm)
            dg *= z - 2 * R * nrm
            g = g * z
        new = np.empty(nc + 1)
        new[0] = v = nodes[0]
        for c in range(nc):                   # node pass: v[c+1] = G[c] + M[c] (v[c] - old[c])
            v = g[c] + dg[c] * (v - nodes[c])
            new[c + 1] = v
        movement = float(np.max(np.abs(new[1:] - nodes[1:]) / np.abs(new[1:])))
        out.append((movement, float(np.max(np.abs(new - truth) / truth)), float(abs(new[-1] - truth[-1]) / truth[-1])))
        nodes = new
    return out


if __name__ == "__main__":
    print("sigma  carrier  start gain   (movement, error, error of the end state) per round")
    for sigma, amp, g0 in [(1e-3, 0.5, 2.0), (1e-2, 0.5, 2.0), (3e-2, 0.5, 2.0), (1e-1, 0.5, 2.0), (1e-2, 0.5, 2.02), (3e-2, 0.5, 1.9)]:
        print(sigma, amp, g0, [tuple(float(f"{v:.3g}") for v in t) for t in rounds(sigma, amp, g0)])
