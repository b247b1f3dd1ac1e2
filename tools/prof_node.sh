#!/bin/bash
# rocprofv3 --kernel-trace --stats of the vertex-model workload tools/prof_node.py (2049^2 x 3, island, 6 RK2 steps)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_node
rm -rf "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o node -- python3 tools/prof_node.py > "$out.log" 2>&1 || { tail -20 "$out.log"; exit 1; }
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/prof_node_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:18]:
    print(f'{r["Name"][:90]:90s} {int(r["Calls"]):6d} {float(r["AverageNs"])/1e3:9.1f} us {float(r["Percentage"]):5.1f} %')
PY
