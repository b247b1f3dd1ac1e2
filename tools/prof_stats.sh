#!/bin/bash
# rocprofv3 --kernel-trace --stats of a bench.py run; prints the top of the per-kernel table and leaves the csv under
# gpurun_out/<tag>/.  usage: tools/prof_stats.sh TAG [bench.py args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$tag" -- python3 bench.py --no-cpu --no-extra "$@" > "$out.json" 2> "$out.err" || { tail -20 "$out.err"; exit 1; }
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
cp "$f" "gpurun_out/${tag}_kernel_stats.csv"
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:24]:
    print(f'{r["Name"][:100]:100s} {int(r["Calls"]):5d} {float(r["AverageNs"])/1e3:9.1f} us {float(r["Percentage"]):5.1f} %')
PY
