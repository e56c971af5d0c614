#!/bin/bash
# usage (GPU box, repo root): tools/ab_env.sh key v1 v2 ...  -- the in-tree build under different values of one switch of the library
# (gz_debug_set through GZ_TEST_SWITCHES: tools/prof_run.py applies it)
# (NDOCS / NO_WORD_TABLE as tools/prof_run.py; REPS default 2)
R=$PWD
cd /tmp && export TMPDIR=/tmp
var=$1; shift
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    tag=${var}_${v}_$rep
    rm -rf /tmp/prof_$tag
    export GZ_TEST_SWITCHES="$var=$v"
    timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof_$tag -o t --output-format csv -- python3 $R/tools/prof_run.py ${NDOCS:-1000000} 5 > /tmp/prof_$tag.log 2>&1 || { tail -20 /tmp/prof_$tag.log; exit 1; }
    echo "== $tag: $(grep '^docs' /tmp/prof_$tag.log | sed 's/.*kernel ms//')"
    python3 $R/tools/trace_summary.py /tmp/prof_$tag | grep -E "classify|words2|mpre|miss|rows|sum of" | awk '{v = $(NF-1); for (i = 1; i < NF; i++) if ($i == "avg") v = $(i+1); printf "%s %s | ", $1, v} END {print ""}'
  done
done
