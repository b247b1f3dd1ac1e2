#!/bin/bash
# rocprofv3 --pmc passes of SQ instruction / cycle counters (separate runs, --kernel-trace only) of a short bench.py run;
# per-kernel means with tools/pmc_sq.py.  usage: tools/prof_sq.sh TAG KERNEL_FILTER [bench.py args...]
set -e
tag=$1; filt=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
files=""
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  out=gpurun_out/${tag}_$n
  rm -rf "$out"
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out" -o p -- python3 bench.py --no-cpu --no-extra --steps 3 --warmup 2 "$@" > "$out.json" 2> "$out.err" || { tail -20 "$out.err"; exit 1; }
  files="$files $(find $out -name '*counter_collection.csv' | head -1)"
done
python3 tools/pmc_sq.py "$filt" $files
