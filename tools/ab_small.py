"""step time at the small reference configurations (C1 128^2 x 1, C2 512^2 x 3, 1024^2 x 3) with option toggles"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, orc
from msom_amd import QG, FIELDS as F
for N, nl in ((128, 1), (512, 3), (1024, 3), (2048, 3)):
    for opt in sys.argv[1:] or ["mg_coarse=0", "mg_coarse=1"]:
        k, v = opt.split("=")
        g = QG(orc.double_gyre_params(N, nl)); g.option("quiet", 1)
        g.set(F["PSI"], orc.synthetic_psi(nl, N, N)); g.set_const(); g.set_tnext(float("inf"))
        g.option(k, float(v))
        for _ in range(5): g.step()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n): g.step()
        g.sync()
        print(f"{N}^2 x {nl}  {opt:16s} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms/step", flush=True)
        g.close()
