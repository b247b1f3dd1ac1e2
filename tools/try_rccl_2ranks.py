"""Probe: two ranks of the RCCL transport on ONE GPU (RCCL normally refuses duplicate devices).
Usage: python tools/try_rccl_2ranks.py            (spawns both ranks)"""
import os, sys, subprocess, time, ctypes
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
if len(sys.argv) == 1:
    uidf = "/tmp/msom_uid.bin"
    if os.path.exists(uidf): os.remove(uidf)
    ps = [subprocess.Popen([sys.executable, __file__, str(r), uidf]) for r in range(2)]
    rc = [p.wait(timeout=100) for p in ps]
    print("exit codes", rc)
    sys.exit(max(rc))
rank, uidf = int(sys.argv[1]), sys.argv[2]
import numpy as np, orc
from msom_amd import QG, FIELDS as F, load_library, tiling
lib = load_library()
if rank == 0:
    uid = tiling.rccl_unique_id(lib)
    open(uidf + ".tmp", "wb").write(uid); os.rename(uidf + ".tmp", uidf)
else:
    while not os.path.exists(uidf): time.sleep(0.05)
    uid = open(uidf, "rb").read()
px, py, tile, nl = 2, 1, 64, 3
params = orc.double_gyre_params(tile * px, nl, extra=f"Ny = {tile}\nMGLEVELS = 6\n")
psi = orc.synthetic_psi(nl, tile, tile * px)
g = QG(params, tiled=(px, py, rank, uid)); g.option("quiet", 1)
g.set(F["PSI"], psi[:, :, rank * tile:(rank + 1) * tile]); g.set_const(); g.set_tnext(float("inf"))
for _ in range(3): g.step()
q = g.get(F["Q"])
np.save(f"/tmp/msom_q{rank}.npy", q)
print("rank", rank, "ok ke", g.ke(), flush=True)
if rank == 0:
    time.sleep(1.0)
    g1 = QG(params); g1.option("quiet", 1); g1.set(F["PSI"], psi); g1.set_const(); g1.set_tnext(float("inf"))
    for _ in range(3): g1.step()
    q0, q1 = np.load("/tmp/msom_q0.npy"), np.load("/tmp/msom_q1.npy")
    print("RCCL 2-rank tiled == single tile:", np.array_equal(np.concatenate([q0, q1], axis=2), g1.get(F["Q"])))
