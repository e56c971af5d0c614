#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc.sh <tag> <n_docs> "<counters pass 1>" "<counters pass 2>" ...
# each pass is its own rocprofv3 run (--pmc only; no trace domains), results under gpurun_out/pmc_<tag>_<k>
tag=$1; ndocs=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
k=0
for pass in "$@"; do
  k=$((k+1))
  rocprofv3 --pmc $pass --output-format csv -d $root/gpurun_out/pmc_${tag}_$k -- python3 $root/tools/prof_run.py $ndocs 2 > $root/gpurun_out/pmc_${tag}_$k.log 2>&1 || exit 1
done
