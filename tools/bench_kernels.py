import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, orc
from msom_amd import QG, FIELDS as F
N, nl = 4096, 6
g = QG(orc.double_gyre_params(N, nl)); g.option("quiet",1)
g.set(F["PSI"], orc.synthetic_psi(nl,N,N)); g.set_const()
for v in range(6):
    g.option("rhs_variant", v)
    print("variant", v, "rhs ms", g.bench_kernel("rhs", 10), flush=True)
print("sweep", g.bench_kernel("sweep", 20), "advance", g.bench_kernel("advance", 10))
