import os, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
from msom_amd import QG, FIELDS as F
N, nl = 32, 3
txt = orc.double_gyre_params(N, nl, extra="afilt = 4\ndtflt = 0.03\n").replace("tend  = 500.", "tend = 0.06").replace("dtout = 1.", "dtout = 0.02")
g = QG(txt); g.option("quiet", 1)
g.set(F["PSI"], orc.synthetic_psi(nl, N, N)); g.set_const()
d = tempfile.mkdtemp()
g.run(d)
od = os.path.join(d, "outdir_0001")
for f in sorted(os.listdir(od)):
    if f.startswith("pf") or f.startswith("po"):
        a = np.fromfile(os.path.join(od, f), "f4").reshape(nl, N + 1, N + 1)[:, 1:, 1:]
        print(f, np.abs(a).max())
print("qof", np.abs(g.get(F["QOF"])).max(), "tmp", np.abs(g.get(F["TMP"])).max())
