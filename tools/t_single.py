#!/usr/bin/env python3
"""Latency of ONE Tokenize.__call__ (the reference's only API) through the GPU path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
from genz_tokenize import Tokenize
sys.path.insert(0, ROOT); import gz_switches; gz_switches.apply()      # GZ_TEST_SWITCHES="key=value,..." -> gz_debug_set (the library reads no switch from the environment)
tok = Tokenize()
cases = [("sinh_viên công_nghệ", "hello", 10), ("xin chào việt nam " * 8, None, 64), ("xin chào việt nam " * 60, None, 256)]
for a, b, L in cases:
    for _ in range(20):
        tok(a, b, max_len=L)
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        tok(a, b, max_len=L)
    dt = (time.perf_counter() - t0) / n
    print("len %4d chars pair=%s max_len=%d: %.1f us per call" % (len(a), b is not None, L, dt * 1e6))
import cProfile, pstats
cProfile.run("for _ in range(200): tok(cases[1][0], None, max_len=64)", "/tmp/p1.out")
pstats.Stats("/tmp/p1.out").sort_stats("cumtime").print_stats(14)
