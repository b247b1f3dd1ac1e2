"""upper bound of what a device-side time loop could save: RK2 step time with the per-solve host read of (max|res|, max|u|) skipped
(option dbg_nosync: the host keeps the last values it read; results meaningless).  usage: python tools/ab_nosync.py N NL"""
import sys, time
sys.path.insert(0, '.')
from msom_amd import QG, FIELDS as F, workloads as wl
N, nl = int(sys.argv[1]), int(sys.argv[2])
g = QG(wl.double_gyre_params(N, nl)); g.option("quiet", 1)
g.set(F["PSI"], wl.synthetic_psi(nl, N, N)); g.set_const(); g.set_tnext(float("inf"))
for _ in range(10): g.step()
n = 200 if N <= 512 else 40
for rep in range(2):
    for ns in (0, 1):
        g.option("dbg_nosync", ns)
        for _ in range(5): g.step()
        t0 = time.perf_counter()
        for _ in range(n): g.step()
        g.sync()
        print(f"N={N} nl={nl} dbg_nosync={ns}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms/step", flush=True)
g.option("dbg_nosync", 0)
