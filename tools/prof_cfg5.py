#!/usr/bin/env python3
"""Driver for rocprofv3 on BASELINE configs[4]: Tokenize.fromFile custom tables (100 000-entry vocab, header-less merges),
50 000 documents of <= 4 000 characters, max_len = 1024 pad + trunc, through the device entry point.  No torch.
usage: prof_cfg5.py [iters]        (RAGGED=1: the unpadded run of the same documents, max_len=None)"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
import numpy as np
import corpus
from genz_tokenize import Tokenize, _native
sys.path.insert(0, ROOT); import gz_switches; gz_switches.apply()      # GZ_TEST_SWITCHES="key=value,..." -> gz_debug_set (the library reads no switch from the environment)

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
v, b = corpus.custom_tables()
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "v"), "wb").write(v); open(os.path.join(tmp, "b"), "wb").write(b)
tok = Tokenize.fromFile(os.path.join(tmp, "v"), os.path.join(tmp, "b")); tok._sync_tables(); ctx = tok._ctx
text, offs, L = corpus.config_corpus(5)
text = np.ascontiguousarray(text); offs = np.ascontiguousarray(offs, dtype=np.int64)
n = len(offs) - 1
d_text = ctx.alloc(len(text) + 64); ctx.h2d(d_text, text)
d_off = ctx.alloc(8 * (n + 1)); ctx.h2d(d_off, offs)
d_ids = ctx.alloc(4 * n * L); d_mask = ctx.alloc(4 * n * L); d_nreal = ctx.alloc(4 * n)
flags = _native.GZ_PADDING | _native.GZ_TRUNCATION | _native.GZ_TIMING
ragged = bool(os.environ.get("RAGGED"))
if ragged:
    cap = len(text) + 2 * n
    ctx.free(d_ids); ctx.free(d_mask)
    d_ids = ctx.alloc(4 * cap); d_mask = ctx.alloc(4 * cap); d_row = ctx.alloc(8 * (n + 1))
    flags |= _native.GZ_MAX_LEN_NONE
ms = []
for _ in range(iters):
    if ragged:
        ctx.encode_device(d_text, d_off, 0, 0, n, 0, flags, cap, d_ids, d_mask, d_row_off=d_row, d_n_real=d_nreal, h_text_off=offs)
    else:
        ctx.encode_device(d_text, d_off, 0, 0, n, L, flags, n * L, d_ids, d_mask, d_n_real=d_nreal)
    ctx.sync()
    ms.append(ctx.timing()[3 if ragged else 0])
nr = np.empty(n, dtype=np.int32); ctx.d2h(nr, d_nreal)
print("docs", n, "bytes", len(text), "tokens", int(nr.sum()), "kernel ms", [round(x, 3) for x in ms],
      "MB/s", round(len(text) / min(ms) / 1e3, 1))
