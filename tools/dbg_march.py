import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, orc
from msom_amd import QG, FIELDS as F
nx,ny,nl=512,128,3
txt=orc.double_gyre_params(nx,nl,extra=f"Ny = {ny}\n")
rng=np.random.default_rng(1)
da0=rng.standard_normal((nl,ny,nx)); res=rng.standard_normal((nl,ny,nx))
out={}
for march in (0,2):
    g=QG(txt,strict=True); g.option("quiet",1); g.option("uniform_S",1); g.option("march",march)
    g.set(F["PSI"],orc.synthetic_psi(nl,ny,nx)); g.set_const()
    for ns in (1,2,3,4):
        out[(march,ns)]=g.relax(0,da0.copy(),res,ns)
    g.close()
for ns in (1,2,3,4):
    d=np.abs(out[(0,ns)]-out[(2,ns)])
    print("nsweeps",ns,"maxdiff",d.max(), "cells",int((d>0).sum()))
    if d.max()>0:
        dm=d.max(axis=0); js,is_=np.where(dm>0); print("  x",is_.min(),is_.max()," y",js.min(),js.max(), " rows", np.unique(js)[:12], " cols", np.unique(is_)[:12])
