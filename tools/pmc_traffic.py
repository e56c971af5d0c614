#!/usr/bin/env python3
"""HBM traffic per kernel and per step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units).
usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [note]"""
import collections, csv, glob, json, sys


def per_kernel(d, counter):
    by = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if k.startswith('gz_') and r['Counter_Name'] == counter:
                by[k].append(float(r['Counter_Value']))
    return {k: sum(v[-2:]) / len(v[-2:]) for k, v in sorted(by.items())}     # the two steady-state launches


fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
step = ('gz_brk_kernel', 'gz_classify_kernel', 'gz_scan32_kernel', 'gz_docw0_kernel', 'gz_words_kernel', 'gz_miss_kernel',
        'gz_miss_wide_kernel', 'gz_long_kernel', 'gz_assemble_kernel')
total = sum(fetch.get(k, 0) + write.get(k, 0) for k in step) * 1024
out = {"workload": "cfg 3: 1 M documents, 287 MB, max_len 256 (tools/prof_run.py 1000000 2)",
       "per_kernel_KiB": {"FETCH_SIZE": fetch, "WRITE_SIZE": write}, "bytes_per_step": int(total),
       "note": (sys.argv[4] if len(sys.argv) > 4 else "") + "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), average of the two "
               "steady-state launches of every kernel; FETCH_SIZE = TCC_EA0_RDREQ x 64 B: reads of wide 16 B/lane streams are "
               "half-counted on gfx950 (the 287 MB input is streamed twice: at most +0.29 GB)"}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print("bytes per step: %.3f GB (read %.3f, written %.3f)" % (total / 1e9, sum(fetch.get(k, 0) for k in step) * 1024 / 1e9,
                                                             sum(write.get(k, 0) for k in step) * 1024 / 1e9))
