#!/usr/bin/env python3
"""Where Tokenize.encode_batch(list of 1 M str, max_len=256) -- timing (iii) of SURVEY.md 8(d) -- spends its time: packing into the
pinned arena, the dense host path of the library (gz_encode_batch: sub-batches, the rows' real entries over the bus, rows padded into
fresh numpy arrays by host threads), and the same call with the host hints off / fewer threads / the dense rows crossing the bus.
usage: t_encode_batch.py [n_docs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
import numpy as np
import corpus
from genz_tokenize import Tokenize, _native, _packing

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tok = Tokenize(); tok._sync_tables(); ctx = tok._ctx
text, offs, L = corpus.config_corpus(3, n_docs=n_docs)
text = np.ascontiguousarray(text); offs = np.ascontiguousarray(offs, dtype=np.int64)
raw = text.tobytes()
docs = [raw[offs[i]:offs[i + 1]].decode("utf-8") for i in range(len(offs) - 1)]
ms = lambda t0: (time.perf_counter() - t0) * 1e3
for rep in range(3):
    t0 = time.perf_counter(); r = tok.encode_batch(docs, max_len=L); a = ms(t0)
    tokens = int(r["attention_mask"].sum(dtype=np.int64)); del r
    t0 = time.perf_counter(); tb, to = _packing.pack_pinned(docs, tok, ctx); b = ms(t0)
    t0 = time.perf_counter(); r = ctx.encode(tb, to, None, None, L, True, True); c = ms(t0); del r
    t0 = time.perf_counter(); r = ctx.encode(text, offs, None, None, L, True, True); d = ms(t0); del r
    print("rep %d: encode_batch(list of str) %.1f ms  |  pack into the pinned arena %.1f  |  gz_encode_batch, pinned text %.1f  |  gz_encode_batch, pageable text %.1f   (%d tokens)" % (rep, a, b, c, d, tokens))
for name, kv in (("host_hints=0 (no MADV_HUGEPAGE / MADV_POPULATE_WRITE)", {"host_hints": 0}), ("host_hints=1 (MADV_HUGEPAGE only)", {"host_hints": 1}),
                 ("host_hints=2 (MADV_POPULATE_WRITE only)", {"host_hints": 2}), ("host_threads=4", {"host_threads": 4}), ("host_threads=8", {"host_threads": 8}),
                 ("host_threads=32", {"host_threads": 32}), ("dense_csr=0 (the dense rows cross the bus, staged through pinned buffers)", {"dense_csr": 0})):
    for k, v in kv.items():
        _native.debug_set(k, v, ctx)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); r = ctx.encode(text, offs, None, None, L, True, True); best = min(best, ms(t0)); del r
    print("gz_encode_batch, pageable text, %-75s %7.1f ms" % (name, best))
    for k in kv:
        _native.debug_set(k, {"host_hints": 3, "host_threads": 0, "dense_csr": 1}[k], ctx)
