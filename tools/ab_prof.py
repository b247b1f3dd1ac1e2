"""profile slots (ms per launch, inside the step) and RK2 step time under option sets, interleaved in one process.
usage: python tools/ab_prof.py N NL "k=v,k=v" "k=v" ...   (an empty string = defaults; NITERMAX = 1 keeps the work fixed)"""
import sys, time
sys.path.insert(0, '.')
from msom_amd import QG, FIELDS as F, workloads as wl
N, nl = int(sys.argv[1]), int(sys.argv[2])
sets = sys.argv[3:] or [""]
g = QG(wl.double_gyre_params(N, nl, extra="NITERMAX = 1\n")); g.option("quiet", 1)
g.set(F["PSI"], wl.synthetic_psi(nl, N, N)); g.set_const(); g.set_tnext(float("inf"))
keys = sorted({kv.split("=")[0] for s in sets for kv in s.split(",") if kv})

SLOTS = ("march_pl", "march_corr", "march4", "march3", "resid_restrict", "resid_max", "rhs")
nst = 10 if N >= 2048 else 50
for _ in range(2): g.step()
for rep in range(2):
    for s in sets:
        cur = dict(kv.split("=") for kv in s.split(",") if kv)
        for k in keys:
            if k in cur: g.option(k, float(cur[k]))
        g.option("profile", 1); g.profile_reset()
        for _ in range(3): g.step()
        g.option("profile", 0)
        prof = " ".join(f"{k}={g.profile_read(k)[0]:.4f}" for k in SLOTS if g.profile_read(k)[0] > 0)
        for _ in range(2): g.step()
        t0 = time.perf_counter()
        for _ in range(nst): g.step()
        g.sync()
        dt = (time.perf_counter() - t0) / nst * 1e3
        print(f"[{s}] step {dt:.4f} ms | {prof}", flush=True)
        for k in keys:   # back to the first set's values (the first set should name every key)
            first = dict(kv.split("=") for kv in sets[0].split(",") if kv)
            if k in first: g.option(k, float(first[k]))
g.close()
