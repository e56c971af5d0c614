#!/usr/bin/env python3
"""Kernels of the last launches of a rocprofv3 --kernel-trace CSV as one timeline ACROSS queues: start / end relative to the first
kernel shown, duration, queue -- and for every kernel which others ran at the same time (overlap in us).  For the rows_apart
experiment: does the row kernel of launch i really run beside the text side of launch i + 1?
usage: trace_overlap.py <trace dir> [launches to show, default 3]"""
import csv, glob, sys
f = (glob.glob(sys.argv[1] + '/*/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*kernel_trace.csv'))[0]
show = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name']
    if name.startswith('void '):
        name = name[5:]
    if name.startswith('gz_'):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name.split('(')[0].split('<')[0], r.get('Queue_Id', '?')))
rows.sort()
cls = [i for i, r in enumerate(rows) if r[2] == 'gz_classify_kernel']
first = cls[-show] if len(cls) >= show else 0
sel = rows[first:]
t0 = sel[0][0]
print("%-24s %5s %9s %9s %8s  %s" % ("kernel", "queue", "start us", "end us", "dur us", "ran beside (us of overlap)"))
for i, (s, e, n, q) in enumerate(sel):
    if n in ('gz_brk_kernel', 'gz_scan32m_kernel', 'gz_scan32_kernel', 'gz_long_kernel'):
        continue
    ov = []
    for j, (s2, e2, n2, q2) in enumerate(sel):
        if j == i or n2 in ('gz_brk_kernel', 'gz_scan32m_kernel', 'gz_scan32_kernel', 'gz_long_kernel', 'gz_docw0_kernel', 'gz_miss_wide_kernel'):
            continue
        o = min(e, e2) - max(s, s2)
        if o > 2000:
            ov.append("%s %.0f" % (n2.replace('gz_', '').replace('_kernel', ''), o / 1e3))
    print("%-24s %5s %9.1f %9.1f %8.1f  %s" % (n.replace('gz_', '').replace('_kernel', ''), q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, ", ".join(ov)))
ends = [e for s, e, n, q in sel if n.startswith('gz_rows1')]
if len(ends) >= 2:
    print("row-kernel end to row-kernel end: %s us" % " ".join("%.1f" % ((b - a) / 1e3) for a, b in zip(ends, ends[1:])))
