"""RK2 step time at 4096^2 x 6 for option SETS, one fresh model per set: python tools/ab_opts.py "a=1,b=0" "c=2" ..."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, orc
from msom_amd import QG, FIELDS as F
N, nl = 4096, 6
psi = orc.synthetic_psi(nl, N, N)
for rep in range(2):
    for spec in [""] + sys.argv[1:]:
        g = QG(orc.double_gyre_params(N, nl)); g.option("quiet", 1)
        g.set(F["PSI"], psi); g.set_const(); g.set_tnext(float("inf"))
        for kv in filter(None, spec.split(",")):
            k, v = kv.split("="); g.option(k, float(v))
        for _ in range(4): g.step()
        t0 = time.perf_counter()
        for _ in range(12): g.step()
        g.sync()
        print(f"{spec or 'default':40s} {(time.perf_counter() - t0) / 12 * 1e3:8.3f} ms/step", flush=True)
        g.close()
