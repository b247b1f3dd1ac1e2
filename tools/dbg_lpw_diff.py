"""does the tendency kernel round the same way in its two instantiations (with / without the ghost-line code)?  One RK2 step with
lpw_dbg = 0 (interior wavefronts take the lean one) against lpw_dbg = 4 (every wavefront takes the one with the wall code)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from msom_amd import QG, FIELDS as F, workloads as wl
N, nl = int(sys.argv[1]), int(sys.argv[2])
out = []
for dbg in (0, 4):
    g = QG(wl.double_gyre_params(N, nl)); g.option("quiet", 1)
    g.set(F["PSI"], wl.synthetic_psi(nl, N, N)); g.set_const(); g.set_tnext(float("inf"))
    g.option("lpw_dbg", dbg)
    g.step()
    out.append(g.get(F["Q"])); g.close()
a, b = out
bad = a != b
print("cells that differ:", bad.sum(), "of", bad.size, " max rel", np.abs(a - b).max() / np.abs(a).max())
if bad.any():
    l, j, i = np.nonzero(bad)
    print("layers", np.bincount(l), "rows", j.min(), j.max(), "cols", i.min(), i.max())
    print("row%4", np.bincount(j % 4), "col%58", np.bincount(i % 58, minlength=58))
