"""a short run of a small configuration for kernel traces (rocprofv3 --kernel-trace -- python3 tools/run_small.py N NL [k=v,...]);
read the trace with tools/trace_step.py"""
import sys
sys.path.insert(0, '.')
from msom_amd import QG, FIELDS as F, workloads as wl
N, nl = int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = QG(wl.double_gyre_params(N, nl)); g.option("quiet", 1)
g.set(F["PSI"], wl.synthetic_psi(nl, N, N)); g.set_const(); g.set_tnext(float("inf"))
for kv in (sys.argv[3].split(",") if len(sys.argv) > 3 else []):
    g.option(kv.split("=")[0], float(kv.split("=")[1]))
for _ in range(20): g.step()
g.close()
