#!/usr/bin/env python3
"""Timing experiment of round 6 (verdict item 5; diagnostic build: GZ_LIBRARY=build_ab/libgz_diag.so): does a word kernel that
classifies its own tiles -- no gz_classify_kernel, no scan of the block counts, no start / end bitmap round trip -- shorten a launch?
Switch `ablate` bit 0 (gz_hot.inc / gz_pipeline.inc).  The block word counts of the call before stay in place, so the rows come out
RIGHT and are compared with the plain run's.  This is the CEILING of the fusion: a real one also needs the global word index of a
block before its records are stored (a look-back inside the persistent kernel), which this run gets for free from the call before.
usage: r06_fused_classify_try.py [n_docs] [reps] [seed]   (1250000 documents + seed 100 = shard 0 of BASELINE configs[3])"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
import numpy as np
import corpus
from genz_tokenize import Tokenize, _native

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 100
tok = Tokenize(); tok._sync_tables(); ctx = tok._ctx
text, offs, L = corpus.config_corpus(4 if seed >= 100 else 3, n_docs=n_docs, seed=seed)
text = np.ascontiguousarray(text); offs = np.ascontiguousarray(offs, dtype=np.int64)
n = len(offs) - 1
d_text = ctx.alloc(len(text) + 64); ctx.h2d(d_text, text)
d_off = ctx.alloc(8 * (n + 1)); ctx.h2d(d_off, offs)
d_ids = ctx.alloc(4 * n * L); d_mask = ctx.alloc(4 * n * L); d_nreal = ctx.alloc(4 * n)
flags = _native.GZ_PADDING | _native.GZ_TRUNCATION | _native.GZ_TIMING


def run(ablate):
    _native.debug_set("ablate", ablate, ctx)
    ctx.encode_device(d_text, d_off, 0, 0, n, L, flags, n * L, d_ids, d_mask, d_n_real=d_nreal, h_text_off=offs)
    ctx.sync()
    return ctx.timing()[0]


run(0); run(0)
want = np.empty((n, L), dtype=np.int32); ctx.d2h(want, d_ids)
res = {0: [], 1: []}
for r in range(reps):
    for a in (0, 1):
        res[a].append(run(a))
        if a == 1 and r == 0:
            got = np.empty((n, L), dtype=np.int32); ctx.d2h(got, d_ids)
            print("rows of the self-classifying run equal the plain run's:", bool(np.array_equal(got, want)))
            del got
for a in (0, 1):
    v = sorted(res[a])
    print("ablate %d: launch ms  min %.4f  median %.4f  all %s" % (a, v[0], v[len(v) // 2], [round(x, 4) for x in res[a]]))
print("difference of the medians: %.1f us per launch of %d documents (%d MB)" % (
    1e3 * (sorted(res[0])[reps // 2] - sorted(res[1])[reps // 2]), n, len(text) // 1000000))
