#!/usr/bin/env python3
"""Diagnostic builds only (make -C genz-tokenize_amd/csrc diag -> build_ab/libgz_diag.so copied over the in-tree library):
runs prof_run's workload and prints the in-kernel stamp sums (gz_diag_prof).  usage: prof_diag.py [n_docs] [iters]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
import numpy as np
import corpus
from genz_tokenize import Tokenize, _native
sys.path.insert(0, ROOT); import gz_switches; gz_switches.apply()      # GZ_TEST_SWITCHES="key=value,..." -> gz_debug_set (the library reads no switch from the environment)

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tok = Tokenize(); tok._sync_tables(); ctx = tok._ctx
lib = _native.load_library()
text, offs, L = corpus.config_corpus(3, n_docs=n_docs)
n = len(offs) - 1
d_text = ctx.alloc(len(text) + 64); ctx.h2d(d_text, text)
d_off = ctx.alloc(8 * (n + 1)); ctx.h2d(d_off, offs)
d_ids = ctx.alloc(4 * n * L); d_mask = ctx.alloc(4 * n * L); d_nreal = ctx.alloc(4 * n)
flags = _native.GZ_PADDING | _native.GZ_TRUNCATION | _native.GZ_TIMING
if os.environ.get("NO_WORD_TABLE"):
    flags |= _native.GZ_NO_WORD_TABLE
buf = (C.c_ulonglong * 64)()
for it in range(iters):
    lib.gz_diag_prof(None, 1)
    ctx.encode_device(d_text, d_off, 0, 0, n, L, flags, n * L, d_ids, d_mask, d_n_real=d_nreal)
    ctx.sync()
    lib.gz_diag_prof(buf, 0)
    print("iter", it, "kernel ms %.3f" % ctx.timing()[0], "prof", [int(x) for x in buf[:12]], "pre", [int(x) for x in buf[12:24]], "rows1", [int(x) for x in buf[24:36]])
