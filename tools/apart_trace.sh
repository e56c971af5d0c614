#!/bin/bash
# usage (GPU box, repo root): tools/apart_trace.sh "<K> <mode>" ...   per-kernel times of tools/apart_try.py under GZ_ROWS_CUS / GZ_ROWS_MASK
set -o pipefail
R=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg
  export GZ_ROWS_CUS=$1 GZ_ROWS_MASK=$2
  rm -rf /tmp/prof_ap
  timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof_ap -o t --output-format csv -- python3 $R/tools/apart_try.py 1000000 8 2 > /tmp/prof_ap.log 2>&1 || { tail -20 /tmp/prof_ap.log; exit 1; }
  echo "== K=$1 mode=$2: $(grep '^GZ_ROWS' /tmp/prof_ap.log)"
  python3 $R/tools/trace_summary.py /tmp/prof_ap | grep -E "classify|scan32|docw0|brk|words|miss|mpre|rows|sum of" | awk '{v = $(NF-1); for (i = 1; i < NF; i++) if ($i == "avg") v = $(i+1); printf "%s %s | ", $1, v} END {print ""}'
done
