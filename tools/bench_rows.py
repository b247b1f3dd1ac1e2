#!/usr/bin/env python
"""Timing of the widened SURVEY 8 rows (a17 vertex variant, N4 wavelet filter and energy budgets) on one
MI355X: wall clock around synchronous C-ABI calls, fields resident in HBM.  Prints one JSON line per row."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402  (parameter templates only)
import orn  # noqa: E402
from msom_amd import FIELDS as F  # noqa: E402
from msom_amd import QG, NodeQG  # noqa: E402


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def main():
    rows = []
    # ---- vertex-grid variant, island mask, no-slip walls
    for N, nl in ((2048, 3), (4096, 3)):
        g = NodeQG(orn.node_params(N, nl, bc_fac=1.0))
        g.set_option("quiet", 1)
        mk = np.ones((1, N + 1, N + 1)); mk[0, N // 4: N // 4 + N // 8, N // 2: N // 2 + N // 8] = 0
        mk[0, 0, :] = mk[0, -1, :] = mk[0, :, 0] = mk[0, :, -1] = 0
        g.set("MASK", mk)
        g.set("PSI", orn.node_psi(nl, N) * mk)
        g.set_const()
        for _ in range(3):
            g.step(True)
        dt = timed(lambda: g.step(True), 10)
        st = g.mgstats()
        rows.append({"row": "a17 vertex variant RK2 step", "grid": f"{N + 1}^2 x {nl}", "ms_per_step": dt * 1e3,
                     "vertex_updates_per_s": (N + 1) ** 2 * nl / dt, "mg_cycles_last_solve": st.i, "resa": st.resa})
        g.close()
    # ---- wavelet filter and energy budgets at the headline size
    N, nl = 4096, 6
    g = QG(orc.double_gyre_params(N, nl, extra="afilt = 0.6\nediag = 0\n"))
    g.option("quiet", 1)
    g.set(F["PSI"], orc.synthetic_psi(nl, N, N))
    g.set_const()
    g.step()
    w = 8.0 * N * N * nl
    dt = timed(lambda: g.wavelet_apply(F["PSI"]), 10)
    bytes_alg = (1.25 + 2.5 + 1.0 / nl) * (4.0 / 3.0) * w
    rows.append({"row": "N4 wavelet transform + scale + inverse (all layers)", "grid": f"{N}^2 x {nl}", "ms": dt * 1e3,
                 "algorithmic_GB": bytes_alg / 1e9, "achieved_GBs": bytes_alg / dt / 1e9})
    dt = timed(lambda: g.wavelet_filter(0.5), 5)
    rows.append({"row": "N4 wavelet_filter (invertq + transform + comp_q + qof)", "grid": f"{N}^2 x {nl}", "ms": dt * 1e3})
    dt = timed(lambda: g.energy_tend(0.01), 5)
    rows.append({"row": "N4 energy_tend (advection_de + dissip_de + ekman_de + running mean)", "grid": f"{N}^2 x {nl}", "ms": dt * 1e3})
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
