#!/bin/bash
# usage (GPU box, repo root): tools/apart_trace.sh "<setting>" ...   kernel timeline ACROSS queues of tools/apart_try.py under each
# setting of the library's switches ("-" = defaults, "rows_apart=1,hot_wgs=256", ...): which kernels overlapped, for how long
set -o pipefail
R=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  rm -rf /tmp/prof_ap
  timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof_ap -o t --output-format csv -- python3 $R/tools/apart_try.py ${NDOCS:-1000000} 8 1 "$cfg" > /tmp/prof_ap.log 2>&1 || { tail -20 /tmp/prof_ap.log; exit 1; }
  echo "== $(grep '^\[' /tmp/prof_ap.log)"
  python3 $R/tools/trace_overlap.py /tmp/prof_ap 3
done
