#!/bin/bash
# usage (GPU box, repo root): tools/pp_trace.sh   -- per-kernel durations of the text pre-pass (tools/prof_pp.py under rocprofv3 --kernel-trace)
set -o pipefail
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp
rocprofv3 --kernel-trace -d /tmp/pp -o t --output-format csv -- python3 $R/tools/prof_pp.py ${NDOCS:-1000000} > /tmp/pp.log 2>&1 || { tail -20 /tmp/pp.log; exit 1; }
grep -E "^all_five|^remove_html" /tmp/pp.log
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pp/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "gz_pp" in r["Kernel_Name"] or "gz_scan" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=len(rows)//6
for tag,chunk in (("all five, last call",rows[2*n:3*n]),("remove_html, last call",rows[5*n:6*n])):
    t0=int(chunk[0]["Start_Timestamp"])
    print("--",tag)
    for r in chunk:
        print("  %-46s %8.1f us   starts at %8.1f" % (r["Kernel_Name"].split("(")[0][:46], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, (int(r["Start_Timestamp"])-t0)/1e3))
PY
