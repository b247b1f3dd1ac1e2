"""timing experiment: back-to-back chained-smoother passes with the split fields addressed [layer][row][x] (the layout) against
[row][layer][x] (dbg_interleave = 1: same buffers, other strides; results meaningless) -- does the distance between the layers of
a row (135 MB at 4096^2) cost bandwidth (TLB reach)?  usage: python tools/ab_interleave.py [N] [nl]"""
import sys
sys.path.insert(0, '.')
from msom_amd import QG, FIELDS as F, workloads as wl
N, nl = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 6
g = QG(wl.double_gyre_params(N, nl)); g.option("quiet", 1)
g.set(F["PSI"], wl.synthetic_psi(nl, N, N)); g.set_const()
g.step()
for rep in range(2):
    for il in (0, 1):
        g.option("dbg_interleave", il)
        for rows in (0, 16, 48):
            g.option("march_rows", rows)
            print(f"interleave={il} rows={rows}", " ".join(f"{k}={g.bench_kernel(k, 10):.4f}" for k in ("march4", "march4p", "march2")), flush=True)
g.option("dbg_interleave", 0)
g.close()
