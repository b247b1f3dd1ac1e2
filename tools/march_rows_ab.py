import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, orc
from msom_amd import QG, FIELDS as F
g = QG(orc.double_gyre_params(4096, 6)); g.option("quiet",1)
g.set(F["PSI"], orc.synthetic_psi(6,4096,4096)); g.set_const()
for rows in [int(a) for a in sys.argv[1:]] * 2:
    g.option("march_rows", rows)
    print("rows", rows, "march4 %.4f march3 %.4f" % (g.bench_kernel("march4", 10), g.bench_kernel("march3", 10)), flush=True)
