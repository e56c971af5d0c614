#!/usr/bin/env python3
"""Idle time on the main queue between consecutive launches of the pipeline in a rocprofv3 --kernel-trace CSV: from the end
of a launch's row kernel to the start of the next launch's classification kernel (back-to-back launches: what the flag
copies, clears and event records between calls cost)."""
import csv, glob, sys
f = (glob.glob(sys.argv[1] + '/*/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*kernel_trace.csv'))[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('void ', '').split('(')[0]
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n))
rows.sort()
last_rows_end = None
gaps = []
for s, e, n in rows:
    if n.startswith('gz_rows1_kernel'):
        last_rows_end = e
    elif n == 'gz_classify_kernel' and last_rows_end is not None:
        gaps.append((s - last_rows_end) / 1e3)
        last_rows_end = None
print("gaps rows1 -> next classify (us):", [round(g, 1) for g in gaps])
between = [(s, e, n) for s, e, n in rows]
# what ran in the last gap
if gaps:
    idx = [i for i, r in enumerate(rows) if r[2].startswith('gz_rows1_kernel')]
    i = idx[int(sys.argv[2])] if len(sys.argv) > 2 else (idx[-2] if len(idx) > 1 else idx[-1])
    print("after that row kernel:")
    for s, e, n in rows[i:i + 10]:
        print("  %-34s start %+9.1f us  dur %7.1f" % (n, (s - rows[i][1]) / 1e3, (e - s) / 1e3))
