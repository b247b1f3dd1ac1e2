"""lean (march_lean = 1) against general (0) body of the chained smoother INSIDE the step: launch times of the profile slots,
with (march_dbg 0) and without (3) memory traffic; then the RK2 step time.  usage: python tools/ab_lean.py [N] [nl]"""
import sys, time
sys.path.insert(0, '.')
from msom_amd import QG, FIELDS as F, workloads as wl
N, nl = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 6
g = QG(wl.double_gyre_params(N, nl, extra="NITERMAX = 1\n")); g.option("quiet", 1)
g.set(F["PSI"], wl.synthetic_psi(nl, N, N)); g.set_const(); g.set_tnext(float("inf"))
for _ in range(2): g.step()
for rep in range(2):
    for lean in (0, 1):
        g.option("march_lean", lean)
        for dbg in (0, 3):
            g.option("march_dbg", dbg); g.option("profile", 1); g.profile_reset()
            for _ in range(3): g.step()
            g.option("profile", 0); g.option("march_dbg", 0)
            print(f"lean={lean} dbg={dbg}", " ".join(f"{k}={g.profile_read(k)[0]:.4f}" for k in ("march_pl", "march_corr", "march4", "resid_max", "rhs")), flush=True)
        for _ in range(3): g.step()
        g.sync() if hasattr(g, "sync") else None
        t0 = time.perf_counter()
        for _ in range(10): g.step()
        print(f"lean={lean} step ms {(time.perf_counter() - t0) / 10 * 1e3:.4f}", flush=True)
g.close()
