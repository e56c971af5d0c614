#!/usr/bin/env python3
"""Construction time of the drop-in (the reference takes 0.27 s for Tokenize(), twice that for fromFile: tokenize.py:39-42,
:261-267), without the table cache, on a cache miss and on a cache hit (gz_cache.inc)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
t0 = time.perf_counter()
from genz_tokenize import Tokenize  # noqa: E402
sys.path.insert(0, ROOT); import gz_switches; gz_switches.apply()      # GZ_TEST_SWITCHES="key=value,..." -> gz_debug_set (the library reads no switch from the environment)
t1 = time.perf_counter()
d = os.path.join(ROOT, "genz-tokenize_amd", "genz_tokenize", "data")


def timed(make):
    a = time.perf_counter()
    tok = make(); tok._sync_tables()
    return tok, time.perf_counter() - a


os.environ["GZ_TABLE_CACHE"] = "off"
os.environ["GZ_LOAD_TIMING"] = "1"           # (read once, at the first load: the library prints the phases of every gz_load_tables to stderr)
tok, first = timed(Tokenize)
print("import %.3f s; first Tokenize() %.3f s (HIP runtime start-up included)" % (t1 - t0, first))
_, nocache = timed(Tokenize)
_, nocache_ff = timed(lambda: Tokenize.fromFile(os.path.join(d, "vocab.txt"), os.path.join(d, "bpe.codes")))
print("no cache:   Tokenize() %.1f ms   fromFile %.1f ms" % (nocache * 1e3, nocache_ff * 1e3))
os.environ["GZ_TABLE_CACHE"] = tempfile.mkdtemp()
m, miss = timed(Tokenize)
h, hit = timed(Tokenize)
h2, hit2 = timed(Tokenize)
ff, hit_ff = timed(lambda: Tokenize.fromFile(os.path.join(d, "vocab.txt"), os.path.join(d, "bpe.codes")))
print("cache miss: Tokenize() %.1f ms (status %d)" % (miss * 1e3, m._ctx.table_cache_status()))
print("cache hit:  Tokenize() %.1f ms, %.1f ms (status %d)   fromFile %.1f ms (status %d)" % (
    hit * 1e3, hit2 * 1e3, h._ctx.table_cache_status(), hit_ff * 1e3, ff._ctx.table_cache_status()))
print("digests equal:", m._ctx.table_digest() == h._ctx.table_digest() == ff._ctx.table_digest())
# where a construction spends its time outside gz_load_tables
from genz_tokenize import _native as N
t = [time.perf_counter()]
ctx = N.Context(None); t.append(time.perf_counter())
v = open(os.path.join(d, "vocab.txt"), encoding="utf-8").read(); b_ = open(os.path.join(d, "bpe.codes"), encoding="utf-8").read(); t.append(time.perf_counter())
vb, bb = v.encode("utf-8"), b_.encode("utf-8"); t.append(time.perf_counter())
os.environ["GZ_TABLE_CACHE"] = "off"
ctx.load_tables(vb, bb, ("<pad>", "<s>", "</s>", "<mask>", "<unk>")); t.append(time.perf_counter())
ctx.decoder_snapshot(); t.append(time.perf_counter())
print("stages: Context() %.1f ms, read + decode both files %.1f ms, encode %.1f ms, load_tables (no cache) %.1f ms, decoder_snapshot %.1f ms" % tuple((t[i + 1] - t[i]) * 1e3 for i in range(5)))
a = time.perf_counter(); _ = h.decoder; b = time.perf_counter()
print("first .decoder access %.1f ms" % ((b - a) * 1e3))
