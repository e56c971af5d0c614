#!/usr/bin/env python3
"""Diagnostic build: which device buffer is READ BEFORE IT IS WRITTEN?  The golden single calls in file order with every fresh
allocation filled with 0xFF / 0xA5 (switch diag_fresh) -- first all of them, then one allocation at a time (diag_fresh_only = k;
the library names the k-th allocation on stderr).  usage: GZ_LIBRARY=build_ab/libgz_diag.so python3 tools/r06_fresh_bisect.py [guard mode]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
from genz_tokenize import Tokenize, _native

rows = [json.loads(l) for l in open(os.path.join(ROOT, "tests", "golden", "g1_cases.jsonl")) if l.strip()]
rows = [r for r in rows if r["kind"] == "call" and "raises" not in r]
guard = int(sys.argv[1]) if len(sys.argv) > 1 else 0


def run(fill, only):
    _native.debug_set("diag_guard", guard)
    _native.debug_set("diag_fresh", fill)
    _native.debug_set("diag_fresh_only", only)
    tok = Tokenize()
    bad = []
    for i, row in enumerate(rows):
        args = [a.encode() if b else a for a, b in zip(row["args"], row.get("bytes_args", [False] * len(row["args"])))]
        try:
            got = json.loads(json.dumps(tok(*args, **row["kwargs"])))
        except Exception as e:  # noqa: BLE001
            got = "raised %r" % (e,)
        if got != row["result"]:
            bad.append(i)
    del tok
    return bad


exact = int(sys.argv[2]) if len(sys.argv) > 2 else 0
_native.debug_set("diag_exact", exact)
print("guard %d exact %d, nothing filled: rows that differ: %s" % (guard, exact, run(0, -1)), flush=True)
if len(sys.argv) > 3:
    sys.exit(0)
for fill in (256, 0xA5 + 1, 1):
    print("guard %d, every fresh allocation filled with 0x%02X: rows that differ: %s" % (guard, fill - 1, run(fill, -1)), flush=True)
for k in range(0, 90):
    sys.stderr.flush()
    bad = run(256, k)
    if bad:
        print("guard %d, ONLY allocation %d filled with 0xFF: rows that differ: %s" % (guard, k, bad), flush=True)
print("done")
