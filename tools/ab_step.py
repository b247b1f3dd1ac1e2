"""A/B timing of one RK2 step under different library options, same process, same GPU."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, orc
from msom_amd import QG, FIELDS as F
N, nl = 4096, 6
g = QG(orc.double_gyre_params(N, nl)); g.option("quiet", 1)
g.set(F["PSI"], orc.synthetic_psi(nl, N, N)); g.set_const(); g.set_tnext(float("inf"))
for _ in range(3): g.step()
def run(n=8):
    t0 = time.perf_counter()
    for _ in range(n): g.step()
    g.sync()
    return (time.perf_counter() - t0) / n * 1e3
cfgs = [("default", {})] + [(a, {a.split('=')[0]: float(a.split('=')[1])}) for a in sys.argv[1:]]
for rep in range(2):
    for name, opts in cfgs:
        for k, v in opts.items(): g.option(k, v)
        run(2)
        print(f"{name:24s} {run():8.3f} ms/step", flush=True)
        for k, v in opts.items():
            if v in (0.0, 1.0): g.option(k, 1.0 - v)   # binary switches go back; other values stay (list them last)
