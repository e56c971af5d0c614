#!/bin/bash
# usage (GPU box, repo root): tools/ab.sh <libA.so> <libB.so> ...   -- A/B the pipeline kernels of several builds on
# the SAME box (boxes differ by a few percent): each build is loaded through GZ_LIBRARY (the product library is never
# touched) and traced REPS times (default 2) in alternating order.  NDOCS / CFG / SEED / NO_WORD_TABLE as tools/prof_run.py.
set -o pipefail
R=$PWD
cd /tmp && export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-2}); do
  for so in "$@"; do
    tag=$(basename $so .so)_$rep
    export GZ_LIBRARY=$R/$so
    rm -rf /tmp/prof_$tag
    timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof_$tag -o t --output-format csv -- python3 $R/tools/prof_run.py ${NDOCS:-1000000} 5 ${CFG:-3} > /tmp/prof_$tag.log 2>&1 || { tail -20 /tmp/prof_$tag.log; exit 1; }
    echo "== $tag: $(grep '^docs' /tmp/prof_$tag.log | sed 's/.*kernel ms//')"
    python3 $R/tools/trace_summary.py /tmp/prof_$tag | grep -E "classify|scan32|docw0|brk|words|miss|mpre|rows|split|assemble|sum of" | awk '{v = $(NF-1); for (i = 1; i < NF; i++) if ($i == "avg") v = $(i+1); printf "%s %s | ", $1, v} END {print ""}'
  done
done
