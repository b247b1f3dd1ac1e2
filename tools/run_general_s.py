"""short run of the metric configuration with a spatially varying Froude field (general column solver) for kernel traces;
usage: rocprofv3 --kernel-trace -d DIR -o NAME -- python3 tools/run_general_s.py [N] [nl] [k=v,...]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from msom_amd import QG, FIELDS as F, workloads as wl
N, nl = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 6
g = QG(wl.double_gyre_params(N, nl)); g.option("quiet", 1)
g.set(F["PSI"], wl.synthetic_psi(nl, N, N))
x = (np.arange(N) + 0.5) / N
shape = 1.0 + 0.3 * np.outer(np.sin(2 * np.pi * x), np.cos(2 * np.pi * x))
g.set(F["FR"], np.stack([g.param(f"Fr_{l}") * shape for l in range(g.shape(F["FR"])[0])]))
g.set_const(); g.set_tnext(float("inf"))
for kv in (sys.argv[3].split(",") if len(sys.argv) > 3 else []):
    g.option(kv.split("=")[0], float(kv.split("=")[1]))
for _ in range(6): g.step()
g.close()
