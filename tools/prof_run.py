#!/usr/bin/env python3
"""Small driver for rocprofv3 (kernel trace or --pmc passes): encodes a cfg-3 corpus a few times through the
device entry point.  No torch.  usage: prof_run.py [n_docs] [iters] [cfg]     (RAGGED=1: without padding, max_len=None)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
import numpy as np
import corpus
from genz_tokenize import Tokenize, _native
sys.path.insert(0, ROOT); import gz_switches; gz_switches.apply()      # GZ_TEST_SWITCHES="key=value,..." -> gz_debug_set (the library reads no switch from the environment)

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tok = Tokenize(); tok._sync_tables(); ctx = tok._ctx
seed = int(os.environ["SEED"]) if os.environ.get("SEED") else None       # (SEED=100, 1250000 documents: shard 0 of BASELINE configs[3])
text, offs, L = corpus.config_corpus(cfg, n_docs=n_docs, seed=seed)
if os.environ.get("TYPOS"):                          # TYPOS=0.2: that fraction of the words replaced by random letters (bench.py's oov_sensitivity)
    text = corpus.add_typos(text, offs, 7, float(os.environ["TYPOS"]))
n = len(offs) - 1
d_text = ctx.alloc(len(text) + 64); ctx.h2d(d_text, text)
d_off = ctx.alloc(8 * (n + 1)); ctx.h2d(d_off, offs)
moff = int(os.environ.get("MASK_OFF", "0"))          # diagnostic: shift the mask array against the ids array
d_ids = ctx.alloc(4 * n * L); d_mask = ctx.alloc(4 * n * L + moff) + moff; d_nreal = ctx.alloc(4 * n)
flags = _native.GZ_PADDING | _native.GZ_TRUNCATION | _native.GZ_TIMING
if os.environ.get("NO_WORD_TABLE"):
    flags |= _native.GZ_NO_WORD_TABLE
ms = []
ragged = bool(os.environ.get("RAGGED"))              # RAGGED=1: the same documents without padding (max_len=None: ragged rows + row offsets)
if ragged:
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    cap = len(text) + 2 * n
    ctx.free(d_ids); ctx.free(d_mask - moff)
    d_ids = ctx.alloc(4 * cap); d_mask = ctx.alloc(4 * cap); d_row = ctx.alloc(8 * (n + 1))
    flags |= _native.GZ_MAX_LEN_NONE
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
