#!/usr/bin/env python3
"""Construction time of the drop-in (the reference takes 0.27 s for Tokenize(), twice that for fromFile)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
t0 = time.perf_counter()
from genz_tokenize import Tokenize  # noqa: E402
t1 = time.perf_counter()
tok = Tokenize(); tok._sync_tables()
t2 = time.perf_counter()
tok2 = Tokenize(); tok2._sync_tables()
t3 = time.perf_counter()
d = os.path.join(ROOT, "genz-tokenize_amd", "genz_tokenize", "data")
tok3 = Tokenize.fromFile(os.path.join(d, "vocab.txt"), os.path.join(d, "bpe.codes")); tok3._sync_tables()
t4 = time.perf_counter()
_ = tok3.decoder
t5 = time.perf_counter()
print("import %.3f s; first Tokenize() + table build %.3f s (HIP runtime start-up included); second %.3f s; fromFile %.3f s; "
      "first .decoder access %.3f s" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
