#!/usr/bin/env python3
"""HBM traffic per kernel and per launch of the pipeline from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB).
usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [pipeline string] [note] [workload]

The LAST launch of the pipeline in each pass is summed: every gz_* dispatch from the last gz_brk_kernel on (the
pipeline's first kernel), so kernels that run more than once per launch count as often as they run."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def kname(r):
    n = r['Kernel_Name']
    if n.startswith('void '):
        n = n[5:]
    return n.split('(')[0]


def last_launch(d, counter):
    rows = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and kname(r).startswith('gz_'):
                rows.append((int(r['Dispatch_Id']), kname(r), float(r['Counter_Value'])))
    rows.sort()
    start = max(i for i, (_, k, _) in enumerate(rows) if k == 'gz_brk_kernel')
    by = collections.OrderedDict()
    for _, k, v in rows[start:]:
        by[k] = by.get(k, 0.0) + v
    return by


fetch, write = last_launch(sys.argv[1], 'FETCH_SIZE'), last_launch(sys.argv[2], 'WRITE_SIZE')
total = (sum(fetch.values()) + sum(write.values())) * 1024
out = {"workload": sys.argv[6] if len(sys.argv) > 6 else "BASELINE configs[2]: 1 M documents, 287 MB, max_len 256 (tools/prof_run.py 1000000 2)",
       "pipeline": sys.argv[4] if len(sys.argv) > 4 else None,
       "source_sha16": __import__("bench").kernel_source_sha16(),      # bench.py quotes this file only for the same kernel sources
       "per_kernel_KiB": {"FETCH_SIZE": fetch, "WRITE_SIZE": write}, "bytes_per_step": int(total),
       "read_bytes": int(sum(fetch.values()) * 1024), "written_bytes": int(sum(write.values()) * 1024),
       "note": (sys.argv[5] if len(sys.argv) > 5 else "") + "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; no trace domains), the last launch "
               "of the pipeline in each pass (every gz_* dispatch from its gz_brk_kernel on); FETCH_SIZE = TCC_EA0_RDREQ x 64 B: reads "
               "of wide 16 B/lane streams are half-counted on gfx950 (MI355X_MICROARCH.md, HBM) -- the 287 MB input is streamed twice "
               "(classify, words), so the true read side is at most 0.29 GB higher; WRITE_SIZE is exact for 16-B-per-lane streaming stores"}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print("bytes per launch: %.3f GB (read %.3f, written %.3f)" % (total / 1e9, sum(fetch.values()) * 1024 / 1e9, sum(write.values()) * 1024 / 1e9))
