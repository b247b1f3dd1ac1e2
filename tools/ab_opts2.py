"""RK2 step time under option sets, interleaved in one process.  usage: python tools/ab_opts2.py N NL "k=v,k=v" "k=v" ... (first set = baseline "")"""
import sys, time
sys.path.insert(0, '.')
from msom_amd import QG, FIELDS as F, workloads as wl
N, nl = int(sys.argv[1]), int(sys.argv[2])
sets = sys.argv[3:]
g = QG(wl.double_gyre_params(N, nl)); g.option("quiet", 1)
g.set(F["PSI"], wl.synthetic_psi(nl, N, N)); g.set_const(); g.set_tnext(float("inf"))
keys = sorted({kv.split("=")[0] for s in sets for kv in s.split(",") if kv})
base = {}
steps = 20 if N >= 2048 else 100
for rep in range(2):
    for s in sets:
        cur = dict(kv.split("=") for kv in s.split(",") if kv)
        for k in keys:
            g.option(k, float(cur.get(k, sys.argv and 0)))
        for _ in range(3): g.step()
        t0 = time.perf_counter()
        for _ in range(steps): g.step()
        g.sync()
        print(f"N={N} nl={nl} [{s}] step ms {(time.perf_counter() - t0) / steps * 1e3:.4f}  cycles {g.mgstats().i}", flush=True)
