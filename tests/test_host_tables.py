"""CPU-only: the host table builder (gz_tables.cpp, through the C ABI's gz_host_tables_*) and the closed-form
pad / pair formulas the kernels use, against the reference's golden vectors and the oracle."""
import base64

import pytest

import gz_oracle as O
from conftest import DATA, read_jsonl
from table_sim import TableSim

native = pytest.importorskip("genz_tokenize._native")


@pytest.fixture(scope="module")
def bundled():
    v = open(DATA + "/vocab.txt", "rb").read()
    b = open(DATA + "/bpe.codes", "rb").read()
    H = native.HostTables(v, b)
    return H, TableSim(H), O.Tables(v, b)


def _cmp_call(sim, row):
    if any(row.get("bytes_args", [])) or not all(isinstance(a, str) for a in row["args"]):
        return 0
    kw = dict(row["kwargs"])
    if kw.pop("return_offset", False):
        return 0
    if "raises" in row:
        if row["raises"] != "ValueError":
            return 0
        with pytest.raises(ValueError):
            sim.call(*row["args"], **kw)
    else:
        got = sim.call(*row["args"], **kw)
        want = {k: v for k, v in row["result"].items() if k != "offset"}
        assert got == want, (row["args"], kw)
    return 1


def test_loader_matches_oracle_dicts(bundled):
    H, _, t = bundled
    assert H.vocab_items() == list(t.encoder.items())          # same insertion order, same ids
    assert H.merge_items() == list(t.ranks.items())
    syms = H.symbols()
    assert len(set(syms)) == len(syms)
    # every symbol's two vocab ids are what the encoder says about its two token spellings
    ids = H.array(2)
    unk = t.unk_id
    for s, (nf, fin) in zip(syms, ids):
        assert nf == t.encoder.get(s + "@@", unk)
        assert fin == (t.encoder.get(s[:-4], unk) if s.endswith("</w>") else unk)


def test_pair_hash_complete(bundled):
    H, sim, t = bundled
    syms = {s: i for i, s in enumerate(H.symbols())}
    n = 0
    for key, r in t.ranks.items():
        if len(key) != 2:
            continue
        a, b = syms[key[0]], syms[key[1]]
        assert sim.probe(a, b) == r
        assert [int(x) for x in sim.merges[r][:3]] == [a, b, syms[key[0] + key[1]]]
        n += 1
    assert n == 50001      # the "#version: 0.2" header is itself a two-field line (rule L6)
    assert sim.probe(syms["n"], syms["n"]) == t.ranks.get(("n", "n"))


def test_perfect_hash_pair_table_is_complete_dense_and_refuses_non_members(bundled):
    """The pair table (8-byte entries, hash and displace, hot set) against the oracle's bpe_ranks: every merge is found -- through the hot set and without it -- with its rank,
    and the merged symbol IS the rank (symbols are numbered by rank; the bundled file has no two lines that spell the same
    string); pairs that are no merges are refused; the table is dense (load > 0.7) and needs no overflow bucket."""
    import random
    H, sim, t = bundled
    syms = {s: i for i, s in enumerate(H.symbols())}
    assert sim.n_overflow == 0 and sim.slots == len(sim.pair8) and sim.nbuckets == len(sim.disp) and sim.nbuckets <= 16384
    n = 0
    for key, r in t.ranks.items():
        if len(key) != 2:
            continue
        a, b = syms[key[0]], syms[key[1]]
        for hot in (True, False):
            assert sim.probe8(a, b, hot) == (r, 0), key
        assert syms[key[0] + key[1]] == r                      # merged symbol == rank
        n += 1
    assert n / sim.slots > 0.7
    rng = random.Random(8)
    ids = list(syms.values())
    members = {(syms[k[0]], syms[k[1]]) for k in t.ranks if len(k) == 2}
    for _ in range(20000):
        a, b = rng.choice(ids), rng.choice(ids)
        if (a, b) not in members:
            assert sim.probe8(a, b) is None and sim.probe(a, b) is None
    assert sim.probe8(0x80000041, 5) is None                    # a code point outside every table never merges


def test_perfect_hash_overflow_buckets_and_alias_flag():
    """Forced overflow buckets (the builder switch ph_force_overflow = 3 -- gz_debug_set: every third bucket is refused, as a table
    the builder cannot place would be) and two merge lines that spell the same string (alias flag: the merged symbol is the
    smaller rank's), in a child process (the switch is process-wide)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r"""
import os, sys
root = %r
for p in (root, os.path.join(root, "tests"), os.path.join(root, "oracle"), os.path.join(root, "genz-tokenize_amd")):
    sys.path.insert(0, p)
from genz_tokenize import _native
from table_sim import TableSim
import gz_oracle as O
_native.debug_set("ph_force_overflow", 3)
for bad in (("no_such_key", 1), ("ph_force_overflow", -1), ("tab_slack", 1), ("small", 2)):      # unknown key / out of range: refused
    try:
        _native.debug_set(*bad); raise SystemExit("gz_debug_set accepted %%r" %% (bad,))
    except ValueError:
        pass
data = os.path.join(root, "genz-tokenize_amd", "genz_tokenize", "data")
v = open(data + "/vocab.txt", "rb").read()
b = open(data + "/bpe.codes", "rb").read()
H = _native.HostTables(v, b); sim = TableSim(H); t = O.Tables(v, b)
assert sim.n_overflow > 1000, sim.n_overflow
syms = {s: i for i, s in enumerate(H.symbols())}
for key, r in t.ranks.items():
    if len(key) == 2:
        assert sim.probe8(syms[key[0]], syms[key[1]], False) == (r, 0), key
for txt in ("sinh_viên công_nghệ zzzqqqxx", "Trường đại_học Công_nghiệp", "a b c"):
    assert sim.call(txt, max_len=16) == O.call(t, txt, max_len=16)
# two lines spell "abc": ("a", "bc") at rank 2 and ("ab", "c") at rank 3 -> one symbol, id 2; rank 3 carries the alias flag
bpe2 = "b c\na b\na bc\nab c\nabc d</w>\n".encode()
voc2 = "abcd 5\nab@@ 4\nabc@@ 3\nd 2\na@@ 1\nbc@@ 1\nc@@ 1\nb@@ 1\n".encode()
H2 = _native.HostTables(voc2, bpe2); s2 = TableSim(H2); t2 = O.Tables(voc2, bpe2)
y = {s: i for i, s in enumerate(H2.symbols())}
assert y["abc"] == 2 and s2.probe8(y["a"], y["bc"], False) == (2, 0) and s2.probe8(y["ab"], y["c"], False) == (3, 1)
assert int(s2.merges[3][2]) == 2
print("ok")
""" % root
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=600, env=dict(os.environ))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-1500:], r.stderr[-3000:])


def test_g1_through_tables(bundled):
    _, sim, _ = bundled
    n = sum(_cmp_call(sim, r) for r in read_jsonl("g1_cases.jsonl") if r["kind"] == "call")
    assert n >= 40


def test_g3_through_tables(bundled):
    _, sim, _ = bundled
    n = sum(_cmp_call(sim, r) for r in read_jsonl("g3_random.jsonl.gz"))
    assert n >= 2000


def test_g4_loader_through_tables():
    for row in read_jsonl("g4_loader.jsonl"):
        v, b = base64.b64decode(row["vocab_b64"]), base64.b64decode(row["bpe_b64"])
        if "calls" not in row:
            if row["raises"] is None:
                native.HostTables(v, b)
            else:
                with pytest.raises(native.GzError) as ei:
                    native.HostTables(v, b)
                assert ei.value.code == native.GZ_E_UTF8
            continue
        H = native.HostTables(v, b)
        assert sorted(H.vocab_items(), key=lambda kv: (kv[1], kv[0])) == [tuple(x) for x in row["encoder"]], row["name"]
        assert sorted(([list(k), r] for k, r in H.merge_items()), key=lambda kv: kv[1]) == row["bpe_ranks"], row["name"]
        sim = TableSim(H)
        for c in row["calls"]:
            _cmp_call(sim, c)
        H.close()


def test_host_builder_under_sanitizers():
    """SURVEY.md section 5 (race detection / sanitizers): the host table builder -- the code that parses untrusted
    vocab / merge file bytes (loader rules L1-L8) and does the index arithmetic behind the device tables -- built with
    g++ -fsanitize=address,undefined (`make -C genz-tokenize_amd/csrc asan`, no HIP involved) and driven through the C
    ABI's gz_host_tables_* in a child process: the bundled tables, every G4 loader fixture, and a byte-level fuzz of
    table files and special-token strings.  Any sanitizer report aborts the child."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    csrc = os.path.join(ROOT, "genz-tokenize_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc, "asan"], check=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not found next to gcc")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    for force in ("0", "3"):                    # the perfect hash as the builder places it, and with every third bucket forced into overflow
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_asan_child.py")], env=dict(env, GZ_TEST_SWITCHES="ph_force_overflow=" + force),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert "asan child ok" in r.stdout
        assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_host_builder_threads_under_tsan(tmp_path):
    """The builder runs the vocab side and the pair table's perfect hash on helper threads: the same sources under
    g++ -fsanitize=thread, over the bundled tables (three builds) and over a vocab file that is not UTF-8 (the helper thread's error
    must come back as the call's).  Any report fails the test."""
    import os
    import subprocess
    from conftest import ROOT, DATA
    csrc = os.path.join(ROOT, "genz-tokenize_amd", "csrc")
    exe = str(tmp_path / "host_tsan")
    c = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-DGZ_HOST_ONLY", "-I", csrc, "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "native", "host_tsan_main.cpp"), os.path.join(csrc, "gz_tables.cpp"), os.path.join(csrc, "gz_host_api.cpp"), "-o", exe, "-lpthread"],
                       capture_output=True, text=True, timeout=600)
    if c.returncode != 0 and "tsan" in (c.stderr or "").lower() and "cannot find" in c.stderr:
        pytest.skip("libtsan is not installed")
    assert c.returncode == 0, c.stderr[-3000:]
    r = subprocess.run([exe, os.path.join(DATA, "vocab.txt"), os.path.join(DATA, "bpe.codes")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66"))
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
    assert "tsan driver ok" in r.stdout and "ThreadSanitizer" not in r.stderr


def test_allocation_failure_in_the_host_builder_is_an_error_code(tmp_path):
    """No C++ exception crosses the C ABI.  tests/native/alloc_fail_main.cpp links the host-only entry points with an operator new
    that throws std::bad_alloc at the k-th allocation and sweeps k over a build of the bundled tables (the first 400 allocations one
    by one, then every 997th of the ~50 000): every failure point -- on the calling thread or on the builder's helper threads -- must
    come back as GZ_E_NOMEM with no tables, never as std::terminate (a silent SIGABRT, rc -6 here).  The GPU-side entry points have the
    same barrier (function-try-blocks, gz_api.cpp) and their own injection test in tests/test_gpu_parity.py."""
    import os
    import subprocess
    from conftest import ROOT, DATA
    csrc = os.path.join(ROOT, "genz-tokenize_amd", "csrc")
    exe = str(tmp_path / "alloc_fail")
    c = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-DGZ_HOST_ONLY", os.path.join(ROOT, "tests", "native", "alloc_fail_main.cpp"),
                        os.path.join(csrc, "gz_tables.cpp"), os.path.join(csrc, "gz_host_api.cpp"), "-o", exe, "-lpthread"],
                       capture_output=True, text=True, timeout=600)
    assert c.returncode == 0, c.stderr[-3000:]
    r = subprocess.run([exe, os.path.join(DATA, "vocab.txt"), os.path.join(DATA, "bpe.codes"), "997"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "rc %d\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "other: 0" in r.stdout and "GZ_E_NOMEM" in r.stdout


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_host_pool_and_row_expansion_under_sanitizers(tmp_path, sanitizer):
    """The HIP-free part of the library's host paths (csrc/gz_hostpool.h: the worker threads of a large host call, the padding of CSR
    rows into the caller's dense arrays, the registry of pinned blocks) built alone with -fsanitize=thread and with
    -fsanitize=address,undefined (tests/native/hostpool_main.cpp): rows padded through the pool the way csr_core uses it -- rotating
    slots, tagged jobs -- equal a restatement of tokenize.py:141-152 for 0 / 1 / 5 threads and both entry widths; the pool's queue,
    parallel(), and its destruction with jobs still queued; the registry under concurrent use.  Any report fails the test."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "hostpool")
    c = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + sanitizer, "-fno-sanitize-recover=undefined",
                        os.path.join(ROOT, "tests", "native", "hostpool_main.cpp"), "-o", exe, "-lpthread"], capture_output=True, text=True, timeout=600)
    if c.returncode != 0 and "cannot find" in (c.stderr or "") and "san" in c.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert c.returncode == 0, c.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66"))
    assert r.returncode == 0 and "host pool driver ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
