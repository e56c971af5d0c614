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
    # A GPU process that dies inside the runtime must leave its last words in the log: errors of the HIP runtime on stderr
    # (AMD_LOG_LEVEL=1: errors only; read when the runtime loads), glibc's "corrupted ..." lines on stderr rather than on a terminal.
    # (The main process too, not only the children of _run_child: round 5's abort came in the main process and pytest's capture
    # kept whatever the runtime had said -- see _gpu_tests_write_to_the_real_stderr below.)
    os.environ.setdefault("AMD_LOG_LEVEL", "1")
    os.environ.setdefault("LIBC_FATAL_STDERR_", "1")
    _install_sigtrace()
    # switches of the library a parent test asked for (GZ_TEST_SWITCHES="small=0,scan_multi=0": gz_switches.py -> gz_debug_set): set
    # as process-wide defaults before any context exists.  The library itself reads no switch from the environment.
    if os.environ.get("GZ_TEST_SWITCHES"):
        import gz_switches
        gz_switches.apply()


SIGTRACE = os.path.join(ROOT, "tests", "native", "sigtrace.so")


def _install_sigtrace():
    """tests/native/sigtrace.c: native frames on stderr when the process dies of SIGABRT / SIGSEGV -- the faulthandler's Python stack
    ends at the ctypes call and cannot say whether the HSA runtime (a GPU memory fault), glibc or std::terminate raised.  Built here
    (gcc, a second), loaded into this process, and named in GZ_SIGTRACE so that `_run_child` preloads it into every child."""
    src = os.path.join(ROOT, "tests", "native", "sigtrace.c")
    try:
        if not os.path.exists(SIGTRACE) or os.path.getmtime(SIGTRACE) < os.path.getmtime(src):
            import subprocess
            subprocess.run(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", SIGTRACE, src], check=True, capture_output=True)
        import ctypes
        ctypes.CDLL(SIGTRACE, mode=ctypes.RTLD_GLOBAL)
        os.environ["GZ_SIGTRACE"] = SIGTRACE
    except Exception as e:  # noqa: BLE001 -- diagnostics must never cost a test run
        sys.stderr.write("[conftest] sigtrace not installed: %s\n" % e)


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


@pytest.fixture(autouse=True)
def _gpu_tests_write_to_the_real_stderr(request):
    """GPU tests run with pytest's fd-level capture suspended: when the process is killed inside a native call (SIGABRT from the
    HIP runtime, glibc or std::terminate) the captured text is never written out, and the one line that names the cause -- "Memory
    access fault ...", "Queue ... aborting with error ...", "free(): invalid pointer", "terminate called after ..." -- is lost.
    Before every such test a line with its name goes to stderr, so that whatever follows in the log belongs to a known test."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    capman = request.config.pluginmanager.getplugin("capturemanager")
    if capman is None:
        yield
        return
    with capman.global_and_fixture_disabled():
        sys.stderr.write("\n[gpu test] %s\n" % request.node.nodeid)
        sys.stderr.flush()
        yield
