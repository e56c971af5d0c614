#!/usr/bin/env python3
"""Driver for rocprofv3: the text pre-pass (all five filters, then remove_html alone) on the 1 M-document corpus, device entry
point, three times each.  usage: prof_pp.py [n_docs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
import numpy as np
import corpus
from genz_tokenize import Tokenize
sys.path.insert(0, ROOT); import gz_switches; gz_switches.apply()      # GZ_TEST_SWITCHES="key=value,..." -> gz_debug_set (the library reads no switch from the environment)

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tok = Tokenize(); tok._sync_tables(); ctx = tok._ctx
text, offs, L = corpus.config_corpus(3, n_docs=n_docs)
n = len(offs) - 1
d_text = ctx.alloc(len(text) + 64); ctx.h2d(d_text, text)
d_off = ctx.alloc(8 * (n + 1)); ctx.h2d(d_off, offs)
d_out = ctx.alloc(len(text) + 64); d_ooff = ctx.alloc(8 * (n + 1))
for name, ops in (("all_five", [1, 2, 3, 4, 5]), ("remove_html", [1])):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        total = ctx.preprocess_device(ops, d_text, d_off, n, len(text), d_out, len(text), d_ooff)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(name, "ms", [round(t, 3) for t in ts], "bytes out", total)
