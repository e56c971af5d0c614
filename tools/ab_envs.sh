#!/bin/bash
# usage (GPU box, repo root): tools/ab_envs.sh "VAR=a VAR2=b" "VAR=c" ...  -- the in-tree build under several environments
# (each argument is a list of assignments, "-" = none), traced once each; REPS=2 repeats the whole series
R=$PWD
cd /tmp && export TMPDIR=/tmp
k=0
for rep in $(seq 1 ${REPS:-1}); do
  for envs in "$@"; do
    k=$((k+1)); tag=env_$k
    rm -rf /tmp/prof_$tag
    [ "$envs" = "-" ] && envs=""
    timeout -k 10 300 env $envs rocprofv3 --kernel-trace -d /tmp/prof_$tag -o t --output-format csv -- python3 $R/tools/prof_run.py ${NDOCS:-1000000} 5 ${CFG:-3} > /tmp/prof_$tag.log 2>&1 || { tail -20 /tmp/prof_$tag.log; exit 1; }
    echo "== [$envs]: $(grep '^docs' /tmp/prof_$tag.log | sed 's/.*kernel ms//')"
    python3 $R/tools/trace_summary.py /tmp/prof_$tag | grep -E "classify|words|miss|mpre|rows|split|assemble|sum of" | awk '{v = $(NF-1); for (i = 1; i < NF; i++) if ($i == "avg") v = $(i+1); printf "%s %s | ", $1, v} END {print ""}'
  done
done
