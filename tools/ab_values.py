"""RK2 step time at 4096^2 x 6 for several values of ONE option: python tools/ab_values.py key v1 v2 ... (same process, same GPU)"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, orc
from msom_amd import QG, FIELDS as F
key, vals = sys.argv[1], [float(v) for v in sys.argv[2:]]
N, nl = 4096, 6
g = QG(orc.double_gyre_params(N, nl)); g.option("quiet", 1)
g.set(F["PSI"], orc.synthetic_psi(nl, N, N)); g.set_const(); g.set_tnext(float("inf"))
for _ in range(3): g.step()
def run(n=8):
    t0 = time.perf_counter()
    for _ in range(n): g.step()
    g.sync()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    for v in vals:
        g.option(key, v); run(2)
        print(f"{key}={v:g}  {run():8.3f} ms/step", flush=True)
