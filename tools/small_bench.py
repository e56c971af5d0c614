import os, sys, time
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
res = []
for k in range(23):
    a = time.perf_counter()
    ctx.encode_device(d_t, d_o, 0, 0, n, L2, flags, n * L2, d_i, d_m, d_n_real=d_r, h_text_off=o2)
    ctx.sync()
    res.append(((time.perf_counter() - a) * 1e3, ctx.timing()[0]))
print("configs[1] %s: wall ms median %.4f  kernels ms median %.4f" % (os.environ.get("GZ_TEST_SWITCHES", "-"), np.median([x for x, _ in res[3:]]), np.median([y for _, y in res[3:]])))

if os.environ.get("SMALL_ONLY"):
    sys.exit(0)


def per_call(fn, reps=2000):
    for _ in range(50):
        fn()
    a = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - a) / reps * 1e6


s1, s2 = "sinh_viên công_nghệ", "hello"
print("single __call__ (text, max_len=10):        %.1f us" % per_call(lambda: tok(s1, max_len=10)))
print("single __call__ (text), ragged:            %.1f us" % per_call(lambda: tok(s1)))
print("encode(text):                              %.1f us" % per_call(lambda: tok.encode(s1, False)))
print("README pair __call__ (a, b, max_len=10):   %.1f us" % per_call(lambda: tok(s1, s2, max_len=10)))
print("pair __call__ (a, b), ragged:              %.1f us" % per_call(lambda: tok(s1, s2)))
