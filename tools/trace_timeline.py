#!/usr/bin/env python3
"""Timeline of the LAST launch of the pipeline in a rocprofv3 --kernel-trace CSV: start (relative to the launch's first kernel),
duration and the idle gap before each kernel -- what the kernel boundaries and cross-stream waits cost."""
import csv, glob, sys
f = (glob.glob(sys.argv[1] + '/*/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*kernel_trace.csv'))[0]
rows = []
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name']
    if name.startswith('void '):
        name = name[5:]
    if name.startswith('gz_'):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name.split('(')[0], r.get('Queue_Id', '?')))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2] == 'gz_brk_kernel']
first = starts[-1]
# the side stream's gz_brk_kernel of this launch may start before the previous launch's last kernel has ended: begin there
t0 = rows[first][0]
end_prev = 0
busy_end = t0
print("%-30s %6s %10s %10s %8s" % ("kernel", "queue", "start us", "dur us", "gap us"))
for s, e, n, q in rows[first:]:
    gap = (s - busy_end) / 1e3
    print("%-30s %6s %10.1f %10.1f %8.1f" % (n, q, (s - t0) / 1e3, (e - s) / 1e3, gap if gap > 0 else 0.0))
    busy_end = max(busy_end, e)
print("launch: first start -> last end %.1f us" % ((busy_end - t0) / 1e3))
