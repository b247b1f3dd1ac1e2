"""per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch), with the gfx950
correction of MI355X_MICROARCH.md (FETCH_SIZE counts the 128-B requests of wide streaming reads at 64 B -> x2).
usage: python tools/pmc_summary.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv"""
import collections
import csv
import json
import sys


def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
N, nl = 4096, 6
wb = 8.0 * N * N * nl
alg = {"k_relax_color_x2<6, true, true>": 1.5, "k_rhs_fused_pipe<32, 512, false>": 3.0, "k_correct_residual<true, true>": 4.0,
       "k_residual2<false, true, true, false>": 3.25 + 1.0 / 16.0, "k_relax_red_prolong3<6, true>": 1.25, "k_relax_march<6, 4, false, false>": 2.5, "k_relax_march<6, 3, false, false>": 2.5, "k_rhs_lpw<4, true, false, true>": 3.0, "k_rhs_lpw<4, true, false, true, false>": 3.0, "k_resmax_march<6>": 2.0,
       "k_relax_march_dma<6, 4, 2, 4, false>": 2.5, "k_relax_march_dma<6, 4, 4, 1, true>": 2.25, "k_relax_march_dma<6, 4, 2, 1, false>": 2.5,
       "k_relax_march_dma<6, 4, 4, 1, true, false>": 1.75, "k_relax_march_dma<6, 4, 2, 4, false, true>": 3.5, "k_relax_march_dma<6, 4, 2, 4, false, false>": 2.5,
       "k_correct_residual<true, false>": 2.0}
rows = []
for k in sorted(f, key=lambda k: -sum(f[k])):
    n = len(f[k])
    multi = "red_prolong" in k or "relax_color_x2<6, true, false>" in k or "k_relax_march" in k or "k_restrict" in k   # multi-level kernels: the finest level
    fk = max(f[k]) if multi else sum(f[k]) / n
    wk = (max(w[k]) if multi else sum(w[k]) / max(1, len(w[k]))) if k in w else 0.0
    row = {"kernel": k, "dispatches": n, "read_GB": fk * 1024 * 2 / 1e9, "write_GB": wk * 1024 / 1e9}
    row["traffic_GB"] = row["read_GB"] + row["write_GB"]
    for a, c in alg.items():
        if a in k:
            row["algorithmic_GB"] = c * wb / 1e9
    rows.append(row)
print(json.dumps(rows[:12], indent=1))
