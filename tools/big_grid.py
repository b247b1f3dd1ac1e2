"""sanity run at a grid larger than the headline one (default 8192^2 x 6): a few RK2 steps, multigrid statistics, KE"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, orc
from msom_amd import QG, FIELDS as F
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 6
g = QG(orc.double_gyre_params(N, nl)); g.option("quiet", 1)
x = (np.arange(N) + 0.5) / N
psi = np.stack([1e-3 * (1 - 0.15 * l) * np.outer(np.sin(np.pi * x) * np.sin(2 * np.pi * x + l), np.sin(np.pi * x)) for l in range(nl)])
g.set(F["PSI"], psi); del psi
g.set_const(); g.set_tnext(float("inf"))
for k in range(4):
    t0 = time.perf_counter(); dt = g.step(); el = time.perf_counter() - t0
    st = g.mgstats()
    print(f"step {k}: dt {dt:.6g}  {el * 1e3:.1f} ms  mg cycles {st.i} resb {st.resb:.3e} resa {st.resa:.3e}  ke {g.ke():.9e}", flush=True)
q = g.get(F["Q"])
print("q finite:", bool(np.isfinite(q).all()), " max|q|", float(np.abs(q).max()))
