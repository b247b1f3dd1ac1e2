"""step time of a px x py tiling on ONE GPU through the in-process transport (tiles as host threads of one process; they
share the GPU, so the wall time of the slowest tile ~ the sum over tiles) under option sets, next to the single-tile run of
the same global grid.  usage: python tools/ab_tiled2.py PX PY TX TY NL ["k=v,k=v" ...]   (first set "" = defaults)"""
import os, sys, threading, time
sys.path.insert(0, '.')
import numpy as np
from msom_amd import QG, FIELDS as F, workloads as wl
px, py, tx, ty, nl = (int(a) for a in sys.argv[1:6])
sets = sys.argv[6:] or [""]
gnx, gny = tx * px, ty * py
params = wl.double_gyre_params(gnx, nl, extra=(f"Ny = {gny}\n" if gny != gnx else ""))
psi = wl.synthetic_psi(nl, gny, gnx)
STEPS = 10
g = QG(params); g.option("quiet", 1); g.set(F["PSI"], psi); g.set_const(); g.set_tnext(float("inf"))
for _ in range(3): g.step()
t0 = time.perf_counter()
for _ in range(STEPS): g.step()
single = (time.perf_counter() - t0) / STEPS * 1e3
print(f"single tile {gnx}x{gny}x{nl}: {single:.3f} ms/step", flush=True)
ref_psi = g.get(F["PSI"])
g.close()
for s in sets:
    opts = dict(kv.split("=") for kv in s.split(",") if kv)
    uid = b"MSOMLOCL" + os.urandom(8) + bytes(112)
    res, out = [None] * (px * py), [None] * (px * py)
    def worker(rank):
        g = QG(params, tiled=(px, py, rank, uid)); g.option("quiet", 1)
        for k, v in opts.items(): g.option(k, float(v))
        ix, iy = rank % px, rank // px
        g.set(F["PSI"], psi[:, iy * ty:(iy + 1) * ty, ix * tx:(ix + 1) * tx]); g.set_const(); g.set_tnext(float("inf"))
        for _ in range(3): g.step()
        t0 = time.perf_counter()
        for _ in range(STEPS): g.step()
        res[rank] = (time.perf_counter() - t0) / STEPS * 1e3
        out[rank] = g.get(F["PSI"])
        g.close()
    th = [threading.Thread(target=worker, args=(r,)) for r in range(px * py)]
    [t.start() for t in th]; [t.join() for t in th]
    full = np.concatenate([np.concatenate([out[iy * px + ix] for ix in range(px)], axis=2) for iy in range(py)], axis=1)
    err = np.abs(full - ref_psi).max() / np.abs(ref_psi).max()
    print(f"[{s}] {max(res):8.3f} ms/step = {max(res) / single:.2f} x single ({px}x{py} tiles of {tx}x{ty}x{nl}); rel diff of psi vs single {err:.1e}", flush=True)
