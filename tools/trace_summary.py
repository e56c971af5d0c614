#!/usr/bin/env python3
"""Per-kernel durations of the steady-state launches in a rocprofv3 --kernel-trace CSV (last 3 launches of each)."""
import collections, csv, glob, sys
f = (glob.glob(sys.argv[1] + '/*/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*kernel_trace.csv'))[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name']
    if name.startswith('void '):                      # template instances: "void gz_rows1_kernel<512>(...)"
        name = name[5:]
    if name.startswith('gz_'):
        by[name.split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0
for k, v in by.items():
    last = v[-3:]
    print("%-24s n=%d last3 avg %8.1f us" % (k, len(v), sum(last) / len(last)))
    if k not in ('gz_scan_kernel', 'gz_finalize_kernel', 'gz_rowlen_kernel'):
        tot += sum(last) / len(last)
print("sum of pipeline kernels %.1f us" % tot)
