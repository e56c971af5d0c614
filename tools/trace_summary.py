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
steps = len(by.get('gz_brk_kernel', [])) or 1        # launches of the pipeline in the trace
tot = 0
for k, v in by.items():
    per = len(v) // steps                            # launches of this kernel per launch of the pipeline (gz_scan32_kernel: 2)
    last = v[-3 * max(per, 1):]
    avg = sum(last) / len(last)
    note = "" if per else "   (not part of every launch: the load-time table build)"
    print("%-24s n=%d last3 avg %8.1f us%s%s" % (k, len(v), avg, " x %d per launch" % per if per > 1 else "", note))
    tot += avg * per
print("sum of pipeline kernels per launch %.1f us" % tot)
