"""vertex model, fixed work (NITERMAX cycles per solve, TOLERANCE 0): step time and finest-level profile slots under option sets.
usage: python tools/ab_node_prof.py N "k=v,k=v" ..."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, orn
from msom_amd import NodeQG
N, nl = int(sys.argv[1]), 3
sets = sys.argv[2:] or [""]
g = NodeQG(orn.node_params(N, nl, bc_fac=1.0)); g.set_option("quiet", 1)
mk = np.ones((1, N + 1, N + 1)); mk[0, N // 4: N // 4 + N // 8, N // 2: N // 2 + N // 8] = 0
mk[0, 0, :] = mk[0, -1, :] = mk[0, :, 0] = mk[0, :, -1] = 0
g.set("MASK", mk); g.set("PSI", orn.node_psi(nl, N) * mk); g.set_const()
g.set_option("NITERMAX", 9); g.set_option("TOLERANCE", 0.)
for _ in range(2): g.step(True)
for rep in range(2):
    for s in sets:
        for kv in s.split(","):
            if kv: g.set_option(kv.split("=")[0], float(kv.split("=")[1]))
        g.step(True)
        t0 = time.perf_counter()
        for _ in range(4): g.step(True)
        dt = (time.perf_counter() - t0) / 4 * 1e3
        g.set_option("profile", 1); g.profile_reset()
        for _ in range(2): g.step(True)
        g.set_option("profile", 0)
        prof = " ".join(f"{k}={g.profile_read(k)[0] * 1e3:.1f}us x{g.profile_read(k)[1] / 2:.0f}" for k in ("relax_fine", "march_fine", "relax_prolong_fine", "residual", "correct", "rhs", "coarse") if g.profile_read(k)[1])
        print(f"[{s}] {dt:.3f} ms/step cycles {g.mgstats().i} | {prof}", flush=True)
