import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, orc
from msom_amd import QG, FIELDS as F
ok = True
for (nx, ny, nl, extra) in [(64,64,3,""),(128,32,6,""),(32,32,1,""),(16,16,2,""),(256,128,4,"sbc = 1.5\nRe = 300\nEks = 0.001\n"),(128,64,2,"")]:
    txt = orc.double_gyre_params(nx, nl, extra=(f"Ny = {ny}\n" if ny!=nx else "") + extra)
    outs = []
    for v in (0, 1):
        g = QG(txt, strict=True); g.option("quiet",1); g.option("rhs_variant", v)
        g.set(F["PSI"], orc.synthetic_psi(nl, ny, nx)); g.set_const()
        g.set(F["QFORC"], 1e-7*np.random.default_rng(3).standard_normal((nl,ny,nx)))
        dq, d = g.update(); 
        for _ in range(2): g.step()
        outs.append((dq, g.get(F["Q"])))
    e = np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    print(nx, ny, nl, "bit-identical:", e); ok &= e
print("ALL OK" if ok else "MISMATCH")
