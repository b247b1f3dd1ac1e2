"""profile slots of the metric configuration with a spatially varying Froude field (general column solver), ms per launch"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from msom_amd import QG, FIELDS as F, workloads as wl
N, nl = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 6
g = QG(wl.double_gyre_params(N, nl)); g.option("quiet", 1)
g.set(F["PSI"], wl.synthetic_psi(nl, N, N))
x = (np.arange(N) + 0.5) / N
shape = 1.0 + 0.3 * np.outer(np.sin(2 * np.pi * x), np.cos(2 * np.pi * x))
g.set(F["FR"], np.stack([g.param(f"Fr_{l}") * shape for l in range(nl - 1)]))
g.set_const(); g.set_tnext(float("inf"))
for kv in sys.argv[3:]:
    g.option(kv.split("=")[0], float(kv.split("=")[1]))
for _ in range(3): g.step()
g.option("profile", 1); g.profile_reset()
for _ in range(5): g.step()
g.option("profile", 0)
for k in ("sweep", "red_prolong", "resid_restrict", "resid_correct", "resid_max", "rhs", "march_pl", "march_corr", "march4", "march3", "march2"):
    ms, n = g.profile_read(k)
    if n: print(f"{k:16s} {ms:.4f} ms x {n / 5:.1f} per step")
t0 = time.perf_counter()
for _ in range(10): g.step()
g.sync()
print("step ms", (time.perf_counter() - t0) * 100, "uniform_S", g.param("uniform_S"), "cycles", g.mgstats().i)
