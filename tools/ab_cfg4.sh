#!/bin/bash
R=$PWD
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for so in "$@"; do
  tag=$(basename $so .so)_$rep
  export GZ_LIBRARY=$R/$so
  rm -rf /tmp/prof_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof_$tag -o t --output-format csv -- python3 $R/tools/prof_cfg5.py 5 > /tmp/prof_$tag.log 2>&1 || { tail -20 /tmp/prof_$tag.log; exit 1; }
  echo "== $tag: $(grep '^docs' /tmp/prof_$tag.log | sed 's/.*kernel ms//')"
  python3 $R/tools/trace_summary.py /tmp/prof_$tag | grep -E "classify|words2|mpre|miss|rows|sum of" | awk '{v = $(NF-1); for (i = 1; i < NF; i++) if ($i == "avg") v = $(i+1); printf "%s %s | ", $1, v} END {print ""}'
done
done
