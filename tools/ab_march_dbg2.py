"""timing decomposition of the two finest-level passes INSIDE the step (march_dbg: 1 = no stores, 2 = no loads, 3 = neither):
results are wrong on purpose, only the durations of the profile slots mean something.  NITERMAX = 1 keeps the work fixed."""
import os, sys
sys.path.insert(0, '.')
from msom_amd import QG, FIELDS as F, workloads as wl
N, nl = int(os.environ.get("N", "4096")), int(os.environ.get("NL", "6"))
for dbg in (0, 1, 2, 3):
    g = QG(wl.double_gyre_params(N, nl, extra="NITERMAX = 1\n")); g.option("quiet", 1)
    g.set(F["PSI"], wl.synthetic_psi(nl, N, N)); g.set_const(); g.set_tnext(float("inf"))
    for _ in range(2): g.step()
    g.option("march_dbg", dbg); g.option("profile", 1); g.profile_reset()
    for _ in range(3): g.step()
    g.option("profile", 0); g.option("march_dbg", 0)
    print(f"dbg={dbg}", " ".join(f"{k}={g.profile_read(k)[0]:.4f}" for k in ("march_pl", "march_corr", "march4")), flush=True)
    g.close()
