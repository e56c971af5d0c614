#!/usr/bin/env python3
"""Experiment: consecutive launches of the pipeline on ONE context (one stream: strictly serial) against launches
alternating between TWO contexts on the same GPU (two streams: the row kernel of one launch, store-bound, can run beside the
classification / word kernels of the next, issue-bound).  usage: overlap_try.py [n_docs] [launches] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
import numpy as np
import corpus
from genz_tokenize import Tokenize, _native
sys.path.insert(0, ROOT); import gz_switches; gz_switches.apply()      # GZ_TEST_SWITCHES="key=value,..." -> gz_debug_set (the library reads no switch from the environment)

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
toks = [Tokenize(), Tokenize()]
for t in toks: t._sync_tables()
ctxs = [t._ctx for t in toks]
text, offs, L = corpus.config_corpus(3, n_docs=n_docs)
n = len(offs) - 1
c0 = ctxs[0]
d_text = c0.alloc(len(text) + 64); c0.h2d(d_text, text)
d_off = c0.alloc(8 * (n + 1)); c0.h2d(d_off, offs)
outs = [(c0.alloc(4 * n * L), c0.alloc(4 * n * L), c0.alloc(4 * n)) for _ in range(2)]
flags = _native.GZ_PADDING | _native.GZ_TRUNCATION | (_native.GZ_TIMING if os.environ.get("TIMING") else 0)
hoff = np.ascontiguousarray(offs, dtype=np.int64)

def run(nctx):
    for c in ctxs: c.sync()
    t0 = time.perf_counter()
    for k in range(launches):
        c = ctxs[k % nctx]; o = outs[k % 2]
        c.encode_device(d_text, d_off, 0, 0, n, L, flags, n * L, o[0], o[1], d_n_real=o[2], h_text_off=hoff)
    for c in ctxs: c.sync()
    return (time.perf_counter() - t0) * 1e3 / launches

for nctx in (1, 2): run(nctx)                      # warm-up (workspaces)
for r in range(reps):
    a = run(1); b = run(2)
    print("rep %d: one context %.3f ms / launch   two contexts alternating %.3f ms / launch   (%.1f %%)" % (r, a, b, 100 * (a - b) / a), flush=True)
ids = [np.empty(n * L, np.int32) for _ in range(2)]
c0.d2h(ids[0], outs[0][0]); c0.d2h(ids[1], outs[1][0])
print("outputs equal:", bool(np.array_equal(ids[0], ids[1])))
