"""one-off drift check of the product build against the CPU oracle over many steps at the reference tolerance (1 cycle per solve):
kinetic energy and q after N steps.  usage: python tools/long_run_check.py [N=256] [nl=3] [steps=400]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import orc
from msom_amd import QG, FIELDS as F
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
txt = orc.double_gyre_params(N, nl)
o = orc.Oracle(txt, smoother=orc.GS_RB, quiet=1)
g = QG(txt); g.option("quiet", 1)
psi = orc.synthetic_psi(nl, N, N)
o.set(orc.PSI, psi); g.set(F["PSI"], psi); o.set_const(); g.set_const()
o.set_tnext(float("inf")); g.set_tnext(float("inf"))
t0 = time.time()
for k in range(1, steps + 1):
    o.step(); g.step()
    if k % 100 == 0 or k == steps:
        qo, qg = o.get(orc.Q), g.get(F["Q"])
        print(f"step {k}: t {g.t:.6f} vs {o.t:.6f}  ke {g.ke():.10e} vs {o.ke():.10e}  rel dq {np.abs(qg - qo).max() / np.abs(qo).max():.3e}  cycles {g.mgstats().i}/{o.mgstats().i}  [{time.time() - t0:.0f} s]", flush=True)
