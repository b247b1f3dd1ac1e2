#!/bin/bash
# two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of a short bench.py run, then the
# per-kernel HBM traffic table (tools/pmc_summary.py).  usage: tools/prof_pmc.sh TAG [bench.py args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/${tag}_$c
  rm -rf "$out"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out" -o p -- python3 bench.py --no-cpu --no-extra --steps 3 --warmup 2 "$@" > "$out.json" 2> "$out.err" || { tail -20 "$out.err"; exit 1; }
done
f=$(find gpurun_out/${tag}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find gpurun_out/${tag}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py "$f" "$w" > gpurun_out/${tag}_traffic.json
cat gpurun_out/${tag}_traffic.json
