import sys, os, threading
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, orc
from msom_amd import QG, FIELDS as F
from test_gpu_tiled import run_tiled, assemble
px, py, tile, nl = 2, 1, 32, 3
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
gnx, gny = tile*px, tile*py
params = orc.double_gyre_params(gnx, nl, extra=(f"Ny = {gny}\n" if gny != gnx else "") + "MGLEVELS = 5\n")
psi = orc.synthetic_psi(nl, gny, gnx)
for pf in (1, 0):
    def fn(g, rank): return None
    # monkeypatch option via env-like: create wrapper
    import test_gpu_tiled as T
    orig = T.QG
    class Q2(orig):
        def __init__(self, *a, **k):
            super().__init__(*a, **k); self.option("prolong_fused", pf)
    T.QG = Q2
    out = run_tiled(params, px, py, psi, nsteps=NS, strict=False)
    T.QG = orig
    g = QG(params, strict=False); g.option("quiet",1); g.option("prolong_fused", pf)
    g.set(F["PSI"], psi); g.set_const(); g.set_tnext(float("inf")); [g.step() for _ in range(NS)]
    for key, fid in (("psi", F["PSI"]), ("q", F["Q"])):
        a, b = assemble(out, key, px, py), g.get(fid)
        d = np.abs(a-b)
        print("prolong_fused", pf, key, "maxdiff", d.max(), "rel", d.max()/np.abs(b).max(), "where", np.unravel_index(d.argmax(), d.shape) if d.max()>0 else None, "ndiff", (d>0).sum())
