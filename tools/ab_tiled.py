"""step time of a px x py tiling run on ONE GPU through the in-process transport (tiles as host threads), under option
toggles: a rough A/B of the tiled code path (the tiles share the GPU, so only the comparison means something).
usage: python tools/ab_tiled.py [tile=2048] [nl=6] key=value ..."""
import os, sys, threading, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, orc
from msom_amd import QG, FIELDS as F
px, py, tile, nl = 2, 1, 2048, 6
cfgs = [("default", {})] + [(a, {a.split('=')[0]: float(a.split('=')[1])}) for a in sys.argv[1:]]
gnx, gny = tile * px, tile * py
params = orc.double_gyre_params(gnx, nl, extra=f"Ny = {gny}\n")
psi = orc.synthetic_psi(nl, gny, gnx)
for name, opts in cfgs:
    uid = b"MSOMLOCL" + os.urandom(8) + bytes(112)
    res = [None] * (px * py)
    def worker(rank):
        g = QG(params, tiled=(px, py, rank, uid)); g.option("quiet", 1)
        for k, v in opts.items(): g.option(k, v)
        ix, iy = rank % px, rank // px
        g.set(F["PSI"], psi[:, iy * tile:(iy + 1) * tile, ix * tile:(ix + 1) * tile]); g.set_const(); g.set_tnext(float("inf"))
        for _ in range(3): g.step()
        t0 = time.perf_counter()
        for _ in range(10): g.step()
        res[rank] = (time.perf_counter() - t0) / 10 * 1e3
        g.close()
    th = [threading.Thread(target=worker, args=(r,)) for r in range(px * py)]
    [t.start() for t in th]; [t.join() for t in th]
    print(f"{name:20s} {max(res):8.3f} ms/step ({px}x{py} tiles of {tile}^2 x {nl} on one GPU)", flush=True)
