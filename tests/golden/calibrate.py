#!/usr/bin/env python3
"""Calibration of the CPU baseline: the REFERENCE's Tokenize.__call__ loop and this repository's
Python restatement (oracle/gz_oracle.py, what bench.py times on the GPU box as `cpu_baseline`, kind "port") on the
same documents, same interpreter, one thread, interleaved.  Writes calibration.json next to this file; bench.py
reports `cpu_baseline.value / port_over_reference` as the reference-equivalent rate.

Run only in the build container (needs /root/reference):  PYTHONDONTWRITEBYTECODE=1 python tests/golden/calibrate.py
"""
import json
import os
import platform
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import corpus  # noqa: E402
import gz_oracle as O  # noqa: E402

sys.path.insert(0, "/root/reference")
from genz_tokenize import Tokenize  # noqa: E402  (the reference)


def cpu_model():
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            return line.split(":", 1)[1].strip()
    return platform.processor()


def main():
    n = 3000
    text, offs, L = corpus.config_corpus(3, n_docs=n)
    raw = text.tobytes()
    docs = [raw[offs[i]:offs[i + 1]].decode("utf-8") for i in range(n)]
    nbytes = int(offs[n])
    ref = Tokenize()
    t = O.Tables(open(corpus.VOCAB_PATH, "rb").read(), open(corpus.BPE_PATH, "rb").read())
    t_ref, t_port = [], []
    for _ in range(3):
        a = time.perf_counter()
        r1 = [ref(d, max_len=L) for d in docs]
        b = time.perf_counter()
        r2 = [O.call(t, d, max_len=L) for d in docs]
        c = time.perf_counter()
        assert all(x["input_ids"] == y["input_ids"] and x["attention_mask"] == y["attention_mask"] for x, y in zip(r1, r2))
        t_ref.append(b - a); t_port.append(c - b)
    ref_mbs, port_mbs = nbytes / min(t_ref) / 1e6, nbytes / min(t_port) / 1e6
    out = {"workload": "first %d documents of BASELINE configs[2] (seed 3, %.2f MB), max_len=%d, one call per document" % (n, nbytes / 1e6, L),
           "reference_MB_per_s": round(ref_mbs, 4), "port_MB_per_s": round(port_mbs, 4),
           "port_over_reference": round(port_mbs / ref_mbs, 4),
           "cpu": cpu_model(), "python": platform.python_version(), "threads": 1,
           "method": "best of 3 interleaved runs, time.perf_counter, outputs compared equal"}
    json.dump(out, open(os.path.join(HERE, "calibration.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
