#!/usr/bin/env python3
"""Diagnostic build, guard-granule allocator: a handful of calls per mode (0 = hipMalloc, 1, 2), results side by side.
usage: GZ_LIBRARY=build_ab/libgz_diag.so python3 tools/r06_guard_debug.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
import numpy as np
import corpus
from genz_tokenize import Tokenize, _native

text, offs, _ = corpus.config_corpus(3, n_docs=3000, seed=5)
text = np.ascontiguousarray(text); offs = np.ascontiguousarray(offs, dtype=np.int64)
ref = {}
for mode in (0, 1, 2):
    _native.debug_set("diag_guard", mode)
    tok = Tokenize()
    out = {}
    out["readme"] = tok('sinh_viên công_nghệ', 'hello', max_len=10)
    out["readme_offsets"] = tok('sinh_viên công_nghệ', 'hello', max_len=10, return_offset=True)
    out["single_offsets"] = tok('sinh_viên công_nghệ xin chào', max_len=12, return_offset=True)
    _native.debug_set("small", 0, tok._ctx)
    out["readme_pipeline"] = tok('sinh_viên công_nghệ', 'hello', max_len=10)
    out["single_pipeline"] = tok('sinh_viên công_nghệ xin chào các bạn', max_len=12)
    _native.debug_set("small", 1, tok._ctx)
    r = tok.encode_packed(text, offs, max_len=64)
    out["batch_3000"] = (int(r["input_ids"].sum(dtype=np.int64)), int(r["attention_mask"].sum(dtype=np.int64)))
    for k, v in out.items():
        if mode == 0:
            ref[k] = v
            print("mode 0", k, v if k != "batch_3000" else v)
        else:
            print("mode %d %-16s %s" % (mode, k, "same as mode 0" if v == ref[k] else "DIFFERS: %s" % (v,)))
    tok._ctx.close() if hasattr(tok._ctx, "close") else None

# ---- the golden single calls in file order under every mode: which rows differ, and do they differ again when repeated at once?
import json
rows = [json.loads(l) for l in open(os.path.join(ROOT, "tests", "golden", "g1_cases.jsonl")) if l.strip()]
for mode in (1, 2, 0):
    _native.debug_set("diag_guard", mode)
    tok = Tokenize()
    bad = []
    for i, row in enumerate(rows):
        if row["kind"] != "call" or "raises" in row:
            continue
        args = [a.encode() if b else a for a, b in zip(row["args"], row.get("bytes_args", [False] * len(row["args"])))]
        got = json.loads(json.dumps(tok(*args, **row["kwargs"])))
        if got != row["result"]:
            again = json.loads(json.dumps(tok(*args, **row["kwargs"])))
            bad.append((i, row["args"], row["kwargs"], "again: %s" % ("same wrong" if again == got else "RIGHT" if again == row["result"] else "other wrong")))
            if len(bad) <= 3:
                print("   got     ", got)
                print("   expected", row["result"])
    print("mode %d: %d call rows, %d differ: %s" % (mode, sum(1 for r in rows if r["kind"] == "call" and "raises" not in r), len(bad), bad[:12]))
