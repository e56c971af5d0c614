#!/usr/bin/env python3
"""Round-5 experiment (verdict item 2): the row kernel of launch i beside the text side of launch i + 1 on the SAME compute units
(option rows_apart: the row kernel on a stream of its own, two workspace slots -- plain streams, no compute-unit masks), with the
text side's persistent kernels sized to leave room (hot_wgs, hot_miss_wgs).  Wall time per launch of back-to-back dense launches
on ONE box, every setting in turn, REPS times in alternating order; a digest of the outputs says they are the same.
usage: apart_try.py [n_docs] [launches] [reps] ["k=v,k=v" ...]      (each argument after the third is one setting; "-" = defaults)"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
import numpy as np
import corpus
import gz_switches
from genz_tokenize import Tokenize, _native

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
settings = sys.argv[4:] or ["-", "rows_apart=1", "rows_apart=1,hot_wgs=256"]
gz_switches.apply()
tok = Tokenize(); tok._sync_tables()
c = tok._ctx
seed = int(os.environ["SEED"]) if os.environ.get("SEED") else None
text, offs, L = corpus.config_corpus(3, n_docs=n_docs, seed=seed)
n = len(offs) - 1
d_text = c.alloc(len(text) + 64); c.h2d(d_text, text)
d_off = c.alloc(8 * (n + 1)); c.h2d(d_off, offs)
outs = [(c.alloc(4 * n * L), c.alloc(4 * n * L), c.alloc(4 * n)) for _ in range(2)]
flags = _native.GZ_PADDING | _native.GZ_TRUNCATION | _native.GZ_TIMING
hoff = np.ascontiguousarray(offs, dtype=np.int64)
DEFAULTS = {"rows_apart": 0, "hot_wgs": 0, "hot_miss_wgs": 0, "side": 1, "brk_side": 1}
if any("rows_prio" in s for s in settings):
    sys.exit("apart_try.py: rows_prio is read when a context makes its rows stream: pass it through GZ_TEST_SWITCHES (process-wide), one run per value")


def run(setting):
    for k, v in DEFAULTS.items():
        _native.debug_set(k, v, c)
    for k, v in gz_switches.parse("" if setting == "-" else setting):
        _native.debug_set(k, v, c)
    c.sync()
    t0 = time.perf_counter()
    for k in range(launches):
        o = outs[k % 2]
        c.encode_device(d_text, d_off, 0, 0, n, L, flags, n * L, o[0], o[1], d_n_real=o[2], h_text_off=hoff)
    hist = c.timing_history(1024)
    wall = (time.perf_counter() - t0) * 1e3 / launches
    h = hashlib.sha256()
    for o in outs:
        for buf in o[:2]:
            a = np.empty(n * L, np.int32); c.d2h(a, buf); h.update(a.tobytes())
    return wall, float(np.mean(hist[1:])) if len(hist) > 1 else float("nan"), h.hexdigest()[:16]


for s in settings:
    run(s)                                                       # warm-up of every setting (streams, workspace slots)
res = {s: [] for s in settings}
for _ in range(reps):
    for s in settings:
        res[s].append(run(s))
for s in settings:
    r = res[s]
    print("[%s]  wall ms/launch %s  min %.3f | by events (launches 2..) %s | digest %s" % (
        s, " ".join("%.3f" % x[0] for x in r), min(x[0] for x in r), " ".join("%.3f" % x[1] for x in r), ",".join(sorted(set(x[2] for x in r)))), flush=True)
