#!/bin/bash
# usage (GPU box, repo root): tools/pmc_ab.sh "<counters>" <libA.so> <libB.so> ...   -- one rocprofv3 --pmc pass (no trace domains) per
# build on tools/prof_run.py (NDOCS documents, default 1 M), builds loaded through GZ_LIBRARY; prints the per-kernel summary
set -o pipefail
R=$PWD
ctr=$1; shift
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  tag=$(basename $so .so)
  export GZ_LIBRARY=$R/$so
  rm -rf /tmp/pmcab_$tag
  timeout -k 10 240 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmcab_$tag -- python3 $R/tools/prof_run.py ${NDOCS:-1000000} 2 ${CFG:-3} > /tmp/pmcab_$tag.log 2>&1 || { tail -5 /tmp/pmcab_$tag.log; exit 1; }
  echo "== $tag"
  python3 $R/tools/pmc_summary.py /tmp/pmcab_$tag | grep -E "^kernel|classify|words|miss2|mpre|rows1"
done
