#!/usr/bin/env python3
"""Where Tokenize.encode_to_device(list of 1 M str) spends its time: packing (by thread count), device allocations, the text's H2D
copy, the kernels, the frees.  usage: t_todevice.py [n_docs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
import numpy as np
import corpus
from genz_tokenize import Tokenize, _native, _packing

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tok = Tokenize(); tok._sync_tables(); ctx = tok._ctx
text, offs, L = corpus.config_corpus(3, n_docs=n_docs)
raw = text.tobytes()
docs = [raw[offs[i]:offs[i + 1]].decode("utf-8") for i in range(len(offs) - 1)]
ms = lambda t0: (time.perf_counter() - t0) * 1e3
for thr in ("8", "16", "32", "64", ""):
    if thr:
        os.environ["GZ_PACK_THREADS"] = thr
    else:
        os.environ.pop("GZ_PACK_THREADS", None)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); t, to = _packing.pack(docs); best = min(best, ms(t0))
    print("pack, GZ_PACK_THREADS=%-3s %7.1f ms" % (thr or "-", best))
n = len(to) - 1
for rep in range(2):
    t0 = time.perf_counter(); d_t = ctx.alloc(t.nbytes + 64); d_o = ctx.alloc(to.nbytes); a = ms(t0)
    t0 = time.perf_counter(); ctx.h2d(d_t, t); ctx.h2d(d_o, to); b = ms(t0)
    t0 = time.perf_counter(); d_i = ctx.alloc(4 * n * L); d_m = ctx.alloc(4 * n * L); d_n = ctx.alloc(4 * n); c = ms(t0)
    t0 = time.perf_counter()
    ctx.encode_device(d_t, d_o, 0, 0, n, L, _native.GZ_PADDING | _native.GZ_TRUNCATION, n * L, d_i, d_m, d_n_real=d_n); ctx.sync(); d = ms(t0)
    nr = np.zeros(n, np.int32)
    t0 = time.perf_counter(); ctx.d2h(nr, d_n); e = ms(t0)
    t0 = time.perf_counter()
    for q in (d_t, d_o, d_i, d_m, d_n):
        ctx.free(q)
    f = ms(t0)
    print("rep %d: alloc text %.1f | h2d text+offsets %.1f | alloc outputs %.1f | encode+sync %.1f | d2h n_real %.1f | frees %.1f ms" % (rep, a, b, c, d, e, f))
for _ in range(3):
    t0 = time.perf_counter(); dv = tok.encode_to_device(docs, max_len=L); print("encode_to_device %.1f ms" % ms(t0)); del dv
