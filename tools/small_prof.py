"""Phase times inside gz_small_kernel (a -DGZ_SMALL_PROF build of the library, e.g. build_ab/small_prof.so copied over the
in-tree one): every workgroup leaves its phase end times (10-ns ticks since its start) in the first cells of its first
row.  usage (GPU box, repo root): python tools/small_prof.py"""
import os, sys
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
import numpy as np, corpus
from genz_tokenize import Tokenize, _native
sys.path.insert(0, ROOT); import gz_switches; gz_switches.apply()      # GZ_TEST_SWITCHES="key=value,..." -> gz_debug_set (the library reads no switch from the environment)
tok = Tokenize(); tok._sync_tables(); ctx = tok._ctx
t2, o2, L2 = corpus.config_corpus(2)
t2 = np.ascontiguousarray(t2); o2 = np.ascontiguousarray(o2, dtype=np.int64)
n = len(o2) - 1
d_t = ctx.alloc(len(t2) + 64); ctx.h2d(d_t, t2)
d_o = ctx.alloc(8 * (n + 1)); ctx.h2d(d_o, o2)
d_i = ctx.alloc(4 * n * L2); d_m = ctx.alloc(4 * n * L2); d_r = ctx.alloc(4 * n)
flags = _native.GZ_PADDING | _native.GZ_TRUNCATION | _native.GZ_TIMING
for k in range(5):
    ctx.encode_device(d_t, d_o, 0, 0, n, L2, flags, n * L2, d_i, d_m, d_n_real=d_r, h_text_off=o2)
    ctx.sync()
ids = np.empty((n, L2), np.int32); ctx.d2h(ids, d_i)
G = int(os.environ.get("G", "14"))
rows = ids[::G, :7].astype(np.float64) / 100.0          # us
names = ["P0 load", "P1 classify", "P2 list", "docw", "P3 words", "P4 merge", "long", "P5 rows"]
print("workgroups %d; kernel time by events %.1f us" % (len(rows), ctx.timing()[0] * 1e3))
print("phase END times since the workgroup's start, us: median / p90 / max")
prev = np.zeros(len(rows))
for i in range(rows.shape[1]):
    c = rows[:, i]
    print("  %-12s end %6.2f / %6.2f / %6.2f    phase itself median %6.2f max %6.2f" % (names[i] if i < len(names) else i, np.median(c), np.percentile(c, 90), c.max(),
                                                                                       np.median(c - prev), (c - prev).max()))
    prev = c
