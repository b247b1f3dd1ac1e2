"""per-pass launch times (profile slots, inside the step) of the prolongation and correction passes under chunk heights
(option march_rows; 0 = automatic).  usage: python tools/ab_march_rows2.py [rows ...]"""
import os, sys
sys.path.insert(0, '.')
from msom_amd import QG, FIELDS as F, workloads as wl
N, nl = int(os.environ.get("N", "4096")), int(os.environ.get("NL", "6"))
g = QG(wl.double_gyre_params(N, nl)); g.option("quiet", 1)
g.set(F["PSI"], wl.synthetic_psi(nl, N, N)); g.set_const(); g.set_tnext(float("inf"))
rows = [int(a) for a in sys.argv[1:]] or [0, 16, 24, 32, 40, 48, 64]
for rep in range(2):
    for r in rows:
        g.option("march_rows", r)
        for _ in range(2): g.step()
        g.option("profile", 1); g.profile_reset()
        for _ in range(5): g.step()
        g.option("profile", 0)
        print(f"rows={r:3d}", " ".join(f"{k}={g.profile_read(k)[0]:.4f}" for k in ("march_pl", "march_corr")), flush=True)
