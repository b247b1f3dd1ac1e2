"""debugging aid: 2 x 2 tiles of 512^2 x 2 (product build) against the single tile, under option sets given as "k=v,k=v" """
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import orc
from msom_amd import QG, FIELDS as F
from test_gpu_tiled import run_tiled, assemble
import os; NS = int(os.environ.get("NS", "3"))
px = py = 2; tile, nl = 512, 2; gn = tile * px
params = orc.double_gyre_params(gn, nl, extra="MGLEVELS = 9\n")
psi = orc.synthetic_psi(nl, gn, gn)
for s in sys.argv[1:] or [""]:
    opts = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in s.split(",") if kv}
    out = run_tiled(params, px, py, psi, nsteps=NS, strict=False, opts=opts)
    g = QG(params, strict=False); g.option("quiet", 1)
    for k, v in opts.items(): g.option(k, v)
    g.set(F["PSI"], psi); g.set_const(); g.set_tnext(float("inf")); [g.step() for _ in range(NS)]
    for key in ("psi", "q"):
        a, b = assemble(out, key, px, py), g.get(F[key.upper()])
        bad = a != b
        print(f"[{s}] {key}: differ {bad.sum()} max rel {np.abs(a - b).max() / np.abs(b).max():.2e}")
        if bad.any():
            l, j, i = np.nonzero(bad)
            print("   layers", np.bincount(l), "rows", j.min(), j.max(), "cols", i.min(), i.max(), "first", list(zip(l[:6], j[:6], i[:6])))
    g.close()
