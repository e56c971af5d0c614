import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "genz-tokenize_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "genz-tokenize_amd", "genz_tokenize", "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_jsonl(name):
    import gzip
    import json
    path = os.path.join(GOLDEN, name)
    op = gzip.open if name.endswith(".gz") else open
    with op(path, "rt", encoding="ascii") as f:
        return [json.loads(l) for l in f if l.strip()]


@pytest.fixture(scope="session")
def oracle_tables():
    import gz_oracle as O
    return O.Tables(open(os.path.join(DATA, "vocab.txt"), "rb").read(),
                    open(os.path.join(DATA, "bpe.codes"), "rb").read())
