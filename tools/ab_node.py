"""vertex variant: step time with option toggles (same process, same GPU).  Usage: python tools/ab_node.py [N [option [values ...]]]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, orn
from msom_amd import NodeQG
N, nl = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 3
g = NodeQG(orn.node_params(N, nl, bc_fac=1.0)); g.set_option("quiet", 1)
mk = np.ones((1, N + 1, N + 1)); mk[0, N // 4: N // 4 + N // 8, N // 2: N // 2 + N // 8] = 0
mk[0, 0, :] = mk[0, -1, :] = mk[0, :, 0] = mk[0, :, -1] = 0
g.set("MASK", mk); g.set("PSI", orn.node_psi(nl, N) * mk); g.set_const()
for _ in range(3): g.step(True)
def run(n=6):
    t0 = time.perf_counter()
    for _ in range(n): g.step(True)
    return (time.perf_counter() - t0) / n * 1e3
import os
for kv in filter(None, os.environ.get("EXTRA", "").split(",")):     # EXTRA=tiled_relax=1,node_march=0: fixed options of the run
    g.set_option(kv.split("=")[0], float(kv.split("=")[1]))
key = sys.argv[2] if len(sys.argv) > 2 else "node_split"
vals = [float(v) for v in sys.argv[3:]] or [0, 129, 65]
for rep in range(2):
    for opt in vals:
        g.set_option(key, opt); run(1)
        print(f"{key}={opt:g}  {run():8.3f} ms/step  cycles {g.mgstats().i}", flush=True)
