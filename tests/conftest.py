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
    # The table cache lives in a directory of this test session unless the caller chose one: tables are then always BUILT by
    # the library under test at least once per session (never read from a user cache another build left behind), and only the
    # dedicated cache test looks at hits.  Child processes of the tests inherit the variable.
    if "GZ_TABLE_CACHE" not in os.environ:
        import atexit
        import shutil
        import tempfile
        d = tempfile.mkdtemp(prefix="gz_table_cache_")
        os.environ["GZ_TABLE_CACHE"] = d
        atexit.register(shutil.rmtree, d, True)
    # switches of the library a parent test asked for (GZ_TEST_SWITCHES="small=0,scan_multi=0": gz_switches.py -> gz_debug_set): set
    # as process-wide defaults before any context exists.  The library itself reads no switch from the environment.
    if os.environ.get("GZ_TEST_SWITCHES"):
        import gz_switches
        gz_switches.apply()


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
