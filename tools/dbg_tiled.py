import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, orc
from msom_amd import QG, FIELDS as F
from test_gpu_tiled import run_tiled, assemble
px,py,tx,ty,nl=2,2,1024,64,4
gnx,gny=tx*px,ty*py
params=orc.double_gyre_params(gnx,nl,extra=f"Ny = {gny}\nMGLEVELS = 6\n")
psi=orc.synthetic_psi(nl,gny,gnx)
for mk in (4,):
    out=run_tiled(params,px,py,psi,nsteps=1,strict=False,opts={"march":2,"march_k":mk,"TOLERANCE":1e30})
    g=QG(params); g.option("quiet",1); g.option("march",2); g.option("march_k",mk); g.option("TOLERANCE",1e30)
    g.set(F["PSI"],psi); g.set_const(); g.set_tnext(float("inf")); g.step()
    a=assemble(out,"psi",px,py); b=g.get(F["PSI"])
    d=np.abs(a-b)
    print("march_k",mk,"cycles",out[0]["st"].i,g.mgstats().i,"maxdiff",d.max(), "rel",d.max()/np.abs(b).max())
    if d.max()>0:
        l,j,i=np.unravel_index(d.argmax(),d.shape); print(" argmax",l,j,i)
        cols=np.where(d.max(axis=(0,1))>1e-3*d.max())[0]; rows=np.where(d.max(axis=(0,2))>1e-3*d.max())[0]
        print(" cols",cols[:10],cols[-10:],len(cols)," rows",rows[:10],len(rows))
    dm=d.max(axis=0)
    js,is_=np.where(dm>0)
    print(" nonzero cells",len(js),"of",dm.size," x range",is_.min() if len(is_) else None,is_.max() if len(is_) else None," y range",js.min() if len(js) else None, js.max() if len(js) else None)
    big=np.where(dm>0.1*dm.max()); print(" big x",np.unique(big[1])[:40])
