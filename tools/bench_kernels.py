import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, orc
from msom_amd import QG, FIELDS as F
N, nl = 4096, int(__import__("os").environ.get("NL", "6"))
g = QG(orc.double_gyre_params(N, nl)); g.option("quiet",1)
g.set(F["PSI"], orc.synthetic_psi(nl,N,N)); g.set_const()
for k in sys.argv[1:] or ["rhs", "sweep", "block2", "block2p", "advance", "residual"]:
    if k.startswith("dbg"):
        g.option("rhs_dbg", int(k[3:])); print("rhs_dbg", k[3:]); continue
    if k.startswith("rv"):
        g.option("rhs_variant", int(k[2:])); print("rhs_variant", k[2:]); continue
    if k.startswith("rr"):
        g.option("rhs_resid", int(k[2:])); print("rhs_resid", k[2:]); continue
    if k.startswith("bv"):
        g.option("block_variant", int(k[2:])); print("block_variant", k[2:]); continue
    print(k, "ms", g.bench_kernel(k, 10), flush=True)
