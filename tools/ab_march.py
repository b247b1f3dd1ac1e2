"""A/B of the chained smoother pass at 4096^2 x 6 (or N, NL from the environment): back-to-back HIP-event timings of
march2/3/4 with the register-window kernel (march_dma=0), the LDS-DMA kernel with one strip (1) or four strips in step (2)
per workgroup, over chunk heights; then the RK2 step time under march_dma x march_prolong, interleaved in one process
(boxes differ by +-10 %).  Usage: python tools/ab_march.py [rows ...]"""
import os
import sys
import time

sys.path.insert(0, '.')
from msom_amd import QG, FIELDS as F, workloads as wl

N, nl = int(os.environ.get("N", "4096")), int(os.environ.get("NL", "6"))
g = QG(wl.double_gyre_params(N, nl))
g.option("quiet", 1)
g.set(F["PSI"], wl.synthetic_psi(nl, N, N))
g.set_const()
rows = [int(a) for a in sys.argv[1:]] or [0, 24, 48]
for dma in (0, 1, 2):
    g.option("march_dma", dma)
    for r in rows:
        g.option("march_rows", r)
        print(f"dma={dma} rows={r:3d}", " ".join(f"{k}={g.bench_kernel(k, 10):.4f}" for k in ("march4", "march3r", "march2")), flush=True)
g.option("march_rows", 0)
g.set_tnext(float("inf"))
for rep in range(2):
    for dma, pl in ((0, 0), (2, 0), (0, 1), (1, 1), (2, 1)):
        g.option("march_dma", dma)
        g.option("march_prolong", pl)
        for _ in range(3):
            g.step()
        t0 = time.perf_counter()
        for _ in range(20):
            g.step()
        g.sync()
        print(f"dma={dma} prolong={pl} step ms {(time.perf_counter() - t0) / 20 * 1e3:.3f}", flush=True)
