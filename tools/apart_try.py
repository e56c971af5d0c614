#!/usr/bin/env python3
"""Experiment (verdict item 4, round 4): wall time per launch of back-to-back dense launches of the pipeline on ONE context, for
the build's GZ_ROWS_CUS setting (the row kernel on its own stream with K compute units, the text side on the other 256 - K).
usage: GZ_ROWS_CUS=K apart_try.py [n_docs] [launches] [reps]     prints ms / launch and a digest of the last two outputs."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
import numpy as np
import corpus
from genz_tokenize import Tokenize, _native

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
tok = Tokenize(); tok._sync_tables()
c = tok._ctx
text, offs, L = corpus.config_corpus(3, n_docs=n_docs)
n = len(offs) - 1
d_text = c.alloc(len(text) + 64); c.h2d(d_text, text)
d_off = c.alloc(8 * (n + 1)); c.h2d(d_off, offs)
outs = [(c.alloc(4 * n * L), c.alloc(4 * n * L), c.alloc(4 * n)) for _ in range(2)]
flags = _native.GZ_PADDING | _native.GZ_TRUNCATION
hoff = np.ascontiguousarray(offs, dtype=np.int64)


def run():
    c.sync()
    t0 = time.perf_counter()
    for k in range(launches):
        o = outs[k % 2]
        c.encode_device(d_text, d_off, 0, 0, n, L, flags, n * L, o[0], o[1], d_n_real=o[2], h_text_off=hoff)
    c.sync()
    return (time.perf_counter() - t0) * 1e3 / launches


run()
ts = [run() for _ in range(reps)]
h = hashlib.sha256()
for o in outs:
    for buf in o[:2]:
        a = np.empty(n * L, np.int32); c.d2h(a, buf); h.update(a.tobytes())
print("GZ_ROWS_CUS=%s  ms/launch %s  min %.3f  digest %s" % (os.environ.get("GZ_ROWS_CUS", "-"), " ".join("%.3f" % t for t in ts), min(ts), h.hexdigest()[:16]), flush=True)
