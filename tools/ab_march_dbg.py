"""timing decomposition of k_relax_march_dma (march_dbg: 1 = no stores, 2 = no loads): results are wrong on purpose,
only the durations mean something"""
import os, sys
sys.path.insert(0, '.')
from msom_amd import QG, FIELDS as F, workloads as wl
N, nl = int(os.environ.get("N", "4096")), int(os.environ.get("NL", "6"))
g = QG(wl.double_gyre_params(N, nl)); g.option("quiet", 1)
g.set(F["PSI"], wl.synthetic_psi(nl, N, N)); g.set_const()
for dma in (1, 2):
    g.option("march_dma", dma)
    for dbg in (0, 1, 2, 3):
        g.option("march_dbg", dbg)
        print(f"dma={dma} dbg={dbg}", " ".join(f"{k}={g.bench_kernel(k, 10):.4f}" for k in ("march4", "march2")), flush=True)
g.option("march_dbg", 0)
print("sweep", g.bench_kernel("sweep", 10), "resid_restrict", g.bench_kernel("resid_restrict", 10), "resid_correct", g.bench_kernel("resid_correct", 10))
