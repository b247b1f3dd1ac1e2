"""where do two option sets differ after ONE multigrid cycle?  (debugging aid)  usage: dbg_lean_diff.py N NL "setA" "setB" """
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from msom_amd import QG, FIELDS as F, workloads as wl
N, nl = int(sys.argv[1]), int(sys.argv[2])
out = []
rng = np.random.default_rng(1)
q = rng.standard_normal((nl, N, N))
for s in sys.argv[3:5]:
    g = QG(wl.double_gyre_params(N, nl)); g.option("quiet", 1); g.set_const()
    g.option("NITERMAX", 1)
    for kv in s.split(","):
        if kv: g.option(kv.split("=")[0], float(kv.split("=")[1]))
    p = np.zeros_like(q)
    g.pyq2p(p, q)
    out.append(p); g.close()
a, b = out
d = np.abs(a - b)
print("max diff", d.max(), "max |a|", np.abs(a).max())
bad = d > 1e-9 * np.abs(a).max()
print("bad cells", bad.sum(), "of", bad.size)
if bad.any():
    l, j, i = np.nonzero(bad)
    print("layers", np.unique(l))
    print("rows: min", j.min(), "max", j.max(), " count by row%14:", np.bincount(j % 14, minlength=14))
    print("first bad rows", np.unique(j)[:40])
    print("cols: min", i.min(), "max", i.max(), " count by (col//2)%60:", np.bincount((i // 2) % 60, minlength=60))
    print("first bad cols", np.unique(i)[:40])
    print("col parity", np.bincount(i % 2), "row+col parity", np.bincount((i + j) % 2))
