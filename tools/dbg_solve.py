"""eigenfunction inverse at full size under option sets: cycles and final residual (debugging aid).
usage: python tools/dbg_solve.py N NL "k=v,k=v" ..."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from msom_amd import QG, workloads as wl
N, nl = int(sys.argv[1]), int(sys.argv[2])
sets = sys.argv[3:] or [""]
g = QG(wl.double_gyre_params(N, nl)); g.option("quiet", 1); g.set_const()
S = [(g.param(f"Fr_{l}") / g.param("Rom")) ** 2 for l in range(nl - 1)]
G = np.zeros((nl, nl))
for l in range(nl):
    if l > 0:
        c = S[l - 1] * g.param(f"idh0_{l}"); G[l, l - 1] += c; G[l, l] -= c
    if l < nl - 1:
        c = S[l] * g.param(f"idh1_{l}"); G[l, l + 1] += c; G[l, l] -= c
gam, vec = np.linalg.eig(G)
D = 80.0 / N
x = (np.arange(N) + 0.5) / N
h = np.outer(np.sin(5 * np.pi * x), np.sin(3 * np.pi * x))
psi = 1e-3 * vec[:, 1].real[:, None, None] * h[None]
lam = -(4 / D**2) * (np.sin(3 * np.pi / (2 * N)) ** 2 + np.sin(5 * np.pi / (2 * N)) ** 2)
q = (lam + gam[1].real) * psi
g.option("TOLERANCE", 1e-7 * np.abs(q).max())
g.option("NITERMAX", 12)
for s in sets:
    for kv in s.split(","):
        if kv: g.option(kv.split("=")[0], float(kv.split("=")[1]))
    p = np.zeros_like(psi)
    g.pyq2p(p, q)
    st = g.mgstats()
    print(f"[{s}] cycles {st.i} resb {st.resb:.3e} resa {st.resa:.3e} nrelax {st.nrelax} err {np.abs(p - psi).max() / np.abs(psi).max():.2e}", flush=True)
