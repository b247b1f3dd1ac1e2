import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, orn
from msom_amd import NodeQG
N, nl = 2048, 3
g = NodeQG(orn.node_params(N, nl, bc_fac=1.0)); g.set_option("quiet", 1); g.set_option("tiled_relax", 0)
mk = np.ones((1, N + 1, N + 1)); mk[0, N // 4: N // 4 + N // 8, N // 2: N // 2 + N // 8] = 0
mk[0, 0, :] = mk[0, -1, :] = mk[0, :, 0] = mk[0, :, -1] = 0
g.set("MASK", mk); g.set("PSI", orn.node_psi(nl, N) * mk); g.set_const()
for _ in range(6): g.step(True)
