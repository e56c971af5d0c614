"""GPU parity: the HIP path (through the C ABI and the drop-in `Tokenize`) against
  - the golden vectors produced by the reference (G1, G3, G4),
  - SHA-256 digests of the reference's outputs on the seeded BASELINE corpora (G5, up to the full 1 M documents),
  - the CPU oracle on hostile / long-word / pair-mode corpora.
Bit-exact everywhere (integer work): every comparison is ==.
"""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

import corpus
import gz_oracle as O
from conftest import DATA, GOLDEN, read_jsonl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tok():
    from genz_tokenize import Tokenize
    return Tokenize()


@pytest.fixture(scope="module")
def sampler():
    return corpus.Sampler()


def _run_child(cmd, name, env=None, timeout=600, cwd=None):
    """Runs a child process of a test and KEEPS what it printed: the whole stdout / stderr go to files -- under
    gpurun_out/children/ (that directory travels back from the GPU box), else in the system's temporary directory -- and on failure
    the assertion message carries the HEAD and the TAIL of both streams (round 4: an abort's first lines -- "Fatal Python error",
    the innermost frame, any HSA or glibc message -- were cut off by a tail-only message).  Python children run with the fault
    handler on; AMD_LOG_LEVEL=1 lets the HIP runtime say which call failed.  Returns the CompletedProcess with .report set."""
    import subprocess
    import sys
    import tempfile
    env = dict(os.environ if env is None else env)
    env.setdefault("PYTHONFAULTHANDLER", "1")
    env.setdefault("AMD_LOG_LEVEL", "1")
    if env.get("GZ_SIGTRACE") and os.path.exists(env["GZ_SIGTRACE"]) and not env.get("LD_PRELOAD"):
        env["LD_PRELOAD"] = env["GZ_SIGTRACE"]                    # native frames when the child dies of a signal (tests/native/sigtrace.c)
    if cmd and cmd[0] == sys.executable and "-X" not in cmd[:3]:
        cmd = [cmd[0], "-X", "faulthandler"] + list(cmd[1:])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "gpurun_out", "children")
    try:
        os.makedirs(d, exist_ok=True)
    except OSError:
        d = tempfile.mkdtemp(prefix="gz_children_")
    safe = "".join(ch if ch.isalnum() or ch in "-_." else "_" for ch in name)
    po, pe = os.path.join(d, safe + ".stdout.txt"), os.path.join(d, safe + ".stderr.txt")
    with open(po, "wb") as fo, open(pe, "wb") as fe:
        try:
            r = subprocess.run(cmd, env=env, cwd=cwd, stdout=fo, stderr=fe, timeout=timeout)
            rc = r.returncode
        except subprocess.TimeoutExpired:
            rc = -999
    out = open(po, "r", errors="replace").read()
    err = open(pe, "r", errors="replace").read()

    def both_ends(t, n=3000):
        return t if len(t) <= 2 * n else t[:n] + "\n[... %d characters left out: %s ...]\n" % (len(t) - 2 * n, "see the file") + t[-n:]
    r = subprocess.CompletedProcess(cmd, rc, out, err)
    r.report = ("child %r: return code %s%s\n---- stdout (%s) ----\n%s\n---- stderr (%s) ----\n%s" %
                (name, rc, " (timed out)" if rc == -999 else "", po, both_ends(out), pe, both_ends(err)))
    return r


def _run_selection(name, env, selection, timeout=600):
    """A selection of this file's GPU tests once more in a child process with other switches in its environment."""
    import sys
    r = _run_child([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider", "-k", selection],
                   name, env=env, timeout=timeout)
    assert r.returncode == 0, r.report
    assert " passed" in r.stdout and "failed" not in r.stdout, r.report
    return r


def _call_matches(tok, row):
    args = list(row["args"])
    for i, isb in enumerate(row.get("bytes_args", [])):
        if isb:
            args[i] = args[i].encode()
    kw = dict(row["kwargs"])
    if "raises" in row:
        exc = {"ValueError": ValueError, "TypeError": TypeError}[row["raises"]]
        with pytest.raises(exc) as ei:
            tok(*args, **kw)
        if row["raises"] == "ValueError":
            assert str(ei.value) == row["msg"]
    else:
        got = tok(*args, **kw)
        assert json.loads(json.dumps(got)) == row["result"], (args, kw)
        assert list(got.keys()) == list(row["result"].keys())
    return 1


def test_g1_cases(tok):
    n = 0
    for row in read_jsonl("g1_cases.jsonl"):
        k = row["kind"]
        if k == "call":
            n += _call_matches(tok, row)
        elif k == "encode":
            got = tok.encode(row["text"], row["return_offset"])
            assert json.loads(json.dumps(got)) == row["result"]
        elif k == "decode":
            assert tok.decode(row["ids"]) == row["result"]
        elif k == "bpe":
            if "raises" in row:
                with pytest.raises(IndexError):
                    tok.bpe(row["word"])
            else:
                assert tok.bpe(row["word"]) == row["result"], row["word"]
        elif k == "vocab_size":
            assert tok.vocab_size() == row["result"]
        elif k == "helpers":
            assert tok.get_atttention_mask(row["ids"]) == row["attention_mask"]
            assert tok.get_sequence_id([1, 5, 2, 2, 6, 2, 0]) == row["sequence_id"]
        elif k in ("get_pairs", "get_pairs_edge"):
            from genz_tokenize.tokenize import get_pairs
            w = row["word"] if row.get("is_str") else tuple(row["word"])
            if "raises" in row:
                with pytest.raises(IndexError):
                    get_pairs(w)
            else:
                assert sorted(list(q) for q in get_pairs(w)) == row["result"]
        elif k == "get_sequence_id":
            assert tok.get_sequence_id(list(row["ids"])) == row["result"]
        elif k == "get_token_type":
            arg = list(row["seq"])
            if "raises" in row:
                with pytest.raises({"ValueError": ValueError, "IndexError": IndexError}[row["raises"]]):
                    tok.get_token_type(arg)
            else:
                assert tok.get_token_type(arg) == row["result"]
            assert arg == row["after"]
        elif k == "get_atttention_mask":
            assert tok.get_atttention_mask(list(row["ids"])) == row["result"]
    assert n >= 45


def test_g3_random_single_calls(tok):
    rows = read_jsonl("g3_random.jsonl.gz")
    n = sum(_call_matches(tok, r) for r in rows)
    assert n >= 2000


def test_g3_random_batched(tok):
    """The same 2 400 calls grouped by keyword arguments and run through the batch API."""
    rows = [r for r in read_jsonl("g3_random.jsonl.gz") if not r["kwargs"].get("return_offset")]
    groups = {}
    for r in rows:
        kw = r["kwargs"]
        key = (len(r["args"]), kw.get("max_len"), kw.get("padding", True), kw.get("truncation", True))
        groups.setdefault(key, []).append(r)
    checked = 0
    for (nargs, ml, pad, tr), rs in groups.items():
        texts = [r["args"][0] for r in rs]
        pairs = [r["args"][1] for r in rs] if nargs == 2 else None
        out = tok.encode_batch(texts, pairs, max_len=ml, padding=pad, truncation=tr)
        ro = out["row_off"]
        ids = out["input_ids"].reshape(-1); mask = out["attention_mask"].reshape(-1)
        for i, r in enumerate(rs):
            a, b = int(ro[i]), int(ro[i + 1])
            if "raises" in r:
                assert out["status"][i] == 1
                continue
            assert out["status"][i] == 0
            assert ids[a:b].tolist() == r["result"]["input_ids"]
            assert mask[a:b].tolist() == r["result"]["attention_mask"]
            if nargs == 2:
                ns, nt = (int(x) for x in out["pair_len"][i])
                seq = out["sequence_id"].reshape(-1)[a:a + ns].tolist()
                tt = out["token_type_ids"].reshape(-1)[a:a + nt].tolist()
                assert [None if v == -1 else v for v in seq] == r["result"]["sequence_id"]
                assert [None if v == -1 else v for v in tt] == r["result"]["token_type_ids"]
            checked += 1
    assert checked >= 1700


def test_g4_loader(tmp_path):
    from genz_tokenize import Tokenize
    for row in read_jsonl("g4_loader.jsonl"):
        vp, bp = tmp_path / (row["name"] + ".vocab"), tmp_path / (row["name"] + ".bpe")
        vp.write_bytes(base64.b64decode(row["vocab_b64"])); bp.write_bytes(base64.b64decode(row["bpe_b64"]))
        if "calls" not in row:
            if row["raises"] is None:
                Tokenize.fromFile(str(vp), str(bp))
            else:
                with pytest.raises(UnicodeDecodeError):
                    Tokenize.fromFile(str(vp), str(bp))
            continue
        t = Tokenize.fromFile(str(vp), str(bp))
        assert sorted(t.encoder.items(), key=lambda kv: (kv[1], kv[0])) == [tuple(x) for x in row["encoder"]], row["name"]
        assert sorted(t.decoder.items()) == [tuple(x) for x in row["decoder"]], row["name"]
        assert sorted(([list(k), r] for k, r in t.bpe_ranks.items()), key=lambda kv: kv[1]) == row["bpe_ranks"]
        assert t.vocab_size() == row["vocab_size"]
        for c in row["calls"]:
            _call_matches(t, c)


def _check_hashes(tok, e, text, offs, word_table=True):
    out = tok.encode_packed(text, offs, max_len=e["max_len"], word_table=word_table)
    ids, mask = out["input_ids"], out["attention_mask"]
    blk = e["block"]
    for k, lo in enumerate(range(0, e["n_docs"], blk)):
        hi = min(e["n_docs"], lo + blk)
        assert hashlib.sha256(ids[lo:hi].tobytes()).hexdigest() == e["ids_sha256"][k], ("ids block", k)
        assert hashlib.sha256(mask[lo:hi].tobytes()).hexdigest() == e["mask_sha256"][k], ("mask block", k)
    assert int(mask.sum()) == e["n_tokens"]
    assert int(out["n_real"].sum()) == e["n_tokens"]


@pytest.mark.parametrize("word_table", [True, False])
@pytest.mark.parametrize("name", ["cfg2_10k", "cfg3_20k", "cfg3_1M"])
def test_g5_reference_hashes_bundled(tok, sampler, name, word_table):
    """BASELINE configs 2 and 3 at FULL size against digests of the reference's own output, with the whole-word
    table and with every word going through the merge loop."""
    e = json.load(open(os.path.join(GOLDEN, "g5_hashes.json")))[name]
    text, offs, L = corpus.config_corpus(e["cfg"], n_docs=e["n_docs"], sampler=sampler)
    assert L == e["max_len"]
    _check_hashes(tok, e, text, offs, word_table)


def test_g5_reference_hashes_custom_tables(sampler, tmp_path):
    """BASELINE config 5: fromFile with 100 k-entry vocab / headerless merges, 4 k-char documents, L=1024."""
    from genz_tokenize import Tokenize
    e = json.load(open(os.path.join(GOLDEN, "g5_hashes.json")))["cfg5_300"]
    v, b = corpus.custom_tables()
    (tmp_path / "v").write_bytes(v); (tmp_path / "b").write_bytes(b)
    t = Tokenize.fromFile(str(tmp_path / "v"), str(tmp_path / "b"))
    text, offs, L = corpus.config_corpus(5, n_docs=e["n_docs"], sampler=sampler)
    _check_hashes(t, e, text, offs, True)
    _check_hashes(t, e, text, offs, False)


def test_cfg4_shard_digests_full_size(tok, sampler):
    """BASELINE configs[3]: one whole shard (1.25 M documents, seed 100 + s) of the fixed 10 M-document job against its
    committed digests (C oracle over the whole shard; the reference itself over the first 20 000 documents).  bench.py
    checks all eight shards on every run; here shard 5."""
    g5 = json.load(open(os.path.join(GOLDEN, "g5_hashes.json")))
    e = g5["cfg4_shard5"]
    text, offs, L = corpus.config_corpus(4, n_docs=e["n_docs"], seed=e["seed"], sampler=sampler)
    assert L == e["max_len"] and int(offs[-1]) == e["input_bytes"]
    _check_hashes(tok, e, text, offs, True)
    r = e["ref_prefix"]
    out = tok.encode_packed(text[:int(offs[r["n_docs"]])], offs[:r["n_docs"] + 1], max_len=L)
    for k, lo in enumerate(range(0, r["n_docs"], r["block"])):
        assert hashlib.sha256(out["input_ids"][lo:lo + r["block"]].tobytes()).hexdigest() == r["ids_sha256"][k]
        assert hashlib.sha256(out["attention_mask"][lo:lo + r["block"]].tobytes()).hexdigest() == r["mask_sha256"][k]


def test_cfg5_full_size_padded_and_unpadded(sampler, tmp_path):
    """BASELINE configs[4] at FULL size: all 50 000 documents (<= 4 000 characters) on the custom 100 k-entry vocabulary,
    max_len=1024 pad+trunc AND unpadded (max_len=None), against the C-oracle digests; the first 300 documents also
    against digests computed by the reference (Tokenize.fromFile) on the same documents."""
    from genz_tokenize import Tokenize
    e = json.load(open(os.path.join(GOLDEN, "g5_hashes.json")))["cfg5_50k"]
    v, b = corpus.custom_tables()
    (tmp_path / "v").write_bytes(v); (tmp_path / "b").write_bytes(b)
    t = Tokenize.fromFile(str(tmp_path / "v"), str(tmp_path / "b"))
    text, offs, L = corpus.config_corpus(5, n_docs=e["n_docs"], sampler=sampler)
    assert int(offs[-1]) == e["input_bytes"]
    for name in ("padded", "unpadded"):
        d = e[name]
        out = t.encode_packed(text, offs, max_len=d["max_len"])
        ids, mask, ro = out["input_ids"].reshape(-1), out["attention_mask"].reshape(-1), out["row_off"]
        for k, lo in enumerate(range(0, e["n_docs"], e["block"])):
            a, z = int(ro[lo]), int(ro[min(lo + e["block"], e["n_docs"])])
            assert hashlib.sha256(ids[a:z].tobytes()).hexdigest() == d["ids_sha256"][k], (name, "ids block", k)
            assert hashlib.sha256(mask[a:z].tobytes()).hexdigest() == d["mask_sha256"][k], (name, "mask block", k)
        assert int(mask.sum()) == d["n_tokens"]
        r = d["ref_prefix"]
        for k, lo in enumerate(range(0, r["n_docs"], r["block"])):
            a, z = int(ro[lo]), int(ro[lo + r["block"]])
            assert hashlib.sha256(ids[a:z].tobytes()).hexdigest() == r["ids_sha256"][k], (name, "reference block", k)
            assert hashlib.sha256(mask[a:z].tobytes()).hexdigest() == r["mask_sha256"][k], (name, "reference block", k)


def test_csr_host_path_matches_dense(tok, sampler, tmp_path):
    """gz_encode_batch_csr (sub-batches, copies overlapped with the kernels, only real entries cross PCIe): the rows it
    returns, padded back on the host, are the dense path's input_ids / attention_mask -- several sub-batches, pinned and
    pageable buffers, empty documents, a 1-document batch, and 32-bit entries for a vocabulary with ids above 65535."""
    from genz_tokenize import Tokenize
    text, offs, L = corpus.config_corpus(3, n_docs=300_000, seed=17, sampler=sampler)       # ~86 MB: 3 sub-batches
    dense = tok.encode_packed(text, offs, max_len=L)
    csr = tok.encode_packed_csr(text, offs, max_len=L)
    assert csr["tokens"].dtype == np.uint16
    ids, mask = tok.csr_to_dense(csr)
    assert np.array_equal(ids, dense["input_ids"]) and np.array_equal(mask, dense["attention_mask"])
    assert np.array_equal(csr["n_real"], dense["n_real"])
    # pinned input and output buffers (the bench's device end-to-end timing uses these)
    ctx = tok._ctx
    ptext = ctx.pinned_empty(len(text), np.uint8); ptext[:] = text
    ptok = ctx.pinned_empty(min(len(offs) * L, len(text) + 2 * len(offs)), np.uint16)
    pnr = ctx.pinned_empty(len(offs) - 1, np.int32)
    t2, n2 = ctx.encode_csr(ptext, offs, L, 16, tokens=ptok, n_real=pnr)
    assert np.array_equal(t2, csr["tokens"]) and np.array_equal(n2, csr["n_real"])
    # ragged edge cases: empty documents, whitespace only, one document, a tiny max_len
    docs = ["", "  ", "a", "", "xin ch\u00e0o c\u00e1c b\u1ea1n", "", "zzzqqqxx " * 40, ""]
    enc = [d.encode() for d in docs]
    o = np.zeros(len(enc) + 1, np.int64); np.cumsum([len(e) for e in enc], out=o[1:])
    t = np.frombuffer(b"".join(enc) + b" ", np.uint8)[:int(o[-1])]
    for ml in (1, 2, 7, 64):
        d = tok.encode_packed(t, o, max_len=ml)
        i2, m2 = tok.csr_to_dense(tok.encode_packed_csr(t, o, max_len=ml))
        assert np.array_equal(i2, d["input_ids"]) and np.array_equal(m2, d["attention_mask"]), ml
    one = tok.encode_packed_csr(t[:0], np.zeros(2, np.int64), max_len=8)
    assert one["tokens"].tolist() == [1, 2] and one["n_real"].tolist() == [2]
    # ids above 65535: 32-bit entries
    v, b = corpus.custom_tables()
    (tmp_path / "v").write_bytes(v); (tmp_path / "b").write_bytes(b)
    t5 = Tokenize.fromFile(str(tmp_path / "v"), str(tmp_path / "b"))
    text5, offs5, L5 = corpus.config_corpus(5, n_docs=200, sampler=sampler)
    c5 = t5.encode_packed_csr(text5, offs5, max_len=L5)
    assert c5["tokens"].dtype == np.int32
    d5 = t5.encode_packed(text5, offs5, max_len=L5)
    i5, m5 = t5.csr_to_dense(c5)
    assert np.array_equal(i5, d5["input_ids"]) and np.array_equal(m5, d5["attention_mask"])


def test_dense_host_path_pads_rows_on_the_host_and_equals_the_device_rows(tok, sampler, tmp_path):
    """gz_encode_batch, large dense single-text calls (the batch form of tokenize.py:184-259): the rows' real entries come over the bus
    in sub-batches and are PADDED INTO THE CALLER'S ARRAYS BY HOST THREADS (csr_core, gz_hostpath.h) -- no caller memory is handed to a
    HIP copy.  Against rows made WITHOUT that machinery: the device entry point's dense rows read back, and the same host call with the
    switch dense_csr = 0 (dense rows over the bus, staged through the library's pinned buffers).  Several sub-batches; pageable and
    pinned text; 1, 3 and the default number of worker threads; long rows of short documents (sub-batches cut by ROWS, not by text);
    a vocabulary whose ids need 32-bit entries; row_off / status / n_real as the header promises them."""
    from genz_tokenize import Tokenize, _native
    ctx = tok._ctx
    text, offs, L = corpus.config_corpus(3, n_docs=300_000, seed=23, sampler=sampler)       # ~86 MB: 3 sub-batches of text
    text = np.ascontiguousarray(text); offs = np.ascontiguousarray(offs, dtype=np.int64)
    n = len(offs) - 1
    # the reference rows: device entry point, rows read back
    d_t = ctx.alloc(len(text) + 64); ctx.h2d(d_t, text)
    d_o = ctx.alloc(8 * (n + 1)); ctx.h2d(d_o, offs)
    d_i, d_m, d_n = ctx.alloc(4 * n * L), ctx.alloc(4 * n * L), ctx.alloc(4 * n)
    ctx.encode_device(d_t, d_o, 0, 0, n, L, _native.GZ_PADDING | _native.GZ_TRUNCATION, n * L, d_i, d_m, d_n_real=d_n, h_text_off=offs)
    ctx.sync()
    want_i = np.empty((n, L), np.int32); want_m = np.empty((n, L), np.int32); want_n = np.empty(n, np.int32)
    ctx.d2h(want_i, d_i); ctx.d2h(want_m, d_m); ctx.d2h(want_n, d_n)
    for q in (d_t, d_o, d_i, d_m, d_n):
        ctx.free(q)
    assert (want_n == L).any() and (want_n < L).any()                # (rows that were cut, rows that were padded)

    def same(r, what):
        assert np.array_equal(r["input_ids"].reshape(n, L), want_i), what
        assert np.array_equal(r["attention_mask"].reshape(n, L), want_m), what
        assert np.array_equal(r["n_real"], want_n) and not r["status"].any(), what
        assert np.array_equal(r["row_off"], np.arange(n + 1, dtype=np.int64) * L), what

    same(ctx.encode(text, offs, None, None, L, True, True), "pageable text")
    ptext = ctx.pinned_empty(len(text), np.uint8); ptext[:] = text
    same(ctx.encode(ptext, offs, None, None, L, True, True), "pinned text")
    for thr in (1, 3):
        _native.debug_set("host_threads", thr, ctx)
        same(ctx.encode(text, offs, None, None, L, True, True), "host_threads=%d" % thr)
    _native.debug_set("host_threads", 0, ctx)
    for hints in (1, 2, 3):
        _native.debug_set("host_hints", hints, ctx)
        same(ctx.encode(text, offs, None, None, L, True, True), "host_hints=%d" % hints)
    _native.debug_set("host_hints", 0, ctx)
    _native.debug_set("dense_csr", 0, ctx)
    same(ctx.encode(text, offs, None, None, L, True, True), "dense_csr=0: the dense rows over the bus")
    _native.debug_set("dense_csr", 1, ctx)
    same(ctx.encode(text, offs, None, None, L, True, True, _native.GZ_NO_WORD_TABLE), "every word through the merge loop")
    # long rows of short documents: 120 000 x 1024 entries = 0.98 GB of rows from 14 MB of text -- sub-batches are cut by rows
    t2, o2, _ = corpus.config_corpus(2, n_docs=120_000, seed=3, sampler=sampler)
    t2 = np.ascontiguousarray(t2); o2 = np.ascontiguousarray(o2, dtype=np.int64)
    r2 = ctx.encode(t2, o2, None, None, 1024, True, True)
    i2, m2 = tok.csr_to_dense(tok.encode_packed_csr(t2, o2, max_len=1024))
    assert np.array_equal(r2["input_ids"].reshape(-1, 1024), i2) and np.array_equal(r2["attention_mask"].reshape(-1, 1024), m2)
    import gz_oracle_c as OC
    co = OC.COracle(open(os.path.join(DATA, "vocab.txt"), "rb").read(), open(os.path.join(DATA, "bpe.codes"), "rb").read())
    k = 2000
    wi, wm, _, _, row, _, _ = co.call_packed(t2[:o2[k]], o2[:k + 1], max_len=1024)
    assert np.array_equal(wi[:int(row[-1])].reshape(k, 1024), r2["input_ids"].reshape(-1, 1024)[:k])
    assert np.array_equal(wm[:int(row[-1])].reshape(k, 1024), r2["attention_mask"].reshape(-1, 1024)[:k])
    del r2, i2, m2
    # ids above 65535: the entries travel as 32-bit words
    v, b = corpus.custom_tables()
    (tmp_path / "v").write_bytes(v); (tmp_path / "b").write_bytes(b)
    t5 = Tokenize.fromFile(str(tmp_path / "v"), str(tmp_path / "b"))
    t5._sync_tables()
    text5, offs5, L5 = corpus.config_corpus(5, n_docs=2500, sampler=sampler)                # 13 MB of text, 20 MB of rows: the large path
    text5 = np.ascontiguousarray(text5); offs5 = np.ascontiguousarray(offs5, dtype=np.int64)
    a = t5._ctx.encode(text5, offs5, None, None, L5, True, True)
    _native.debug_set("dense_csr", 0, t5._ctx)
    b5 = t5._ctx.encode(text5, offs5, None, None, L5, True, True)
    assert int(a["input_ids"].max()) > 65535
    assert np.array_equal(a["input_ids"], b5["input_ids"]) and np.array_equal(a["attention_mask"], b5["attention_mask"]) and np.array_equal(a["n_real"], b5["n_real"])


def test_big_pipeline_on_small_inputs():
    """Small dense single-text batches run in ONE fused launch (gz_small_kernel); the switch small = 0 sends them through the
    kernel pipeline instead.  The golden vectors and the small-input comparisons of this file run again that way in a
    child process, so that both forms stay pinned to the reference on the hostile small cases."""
    # (scan_multi = 0: the block-count scans go through the chained multi-workgroup kernel whatever the size -- large
    # batches use it by default, here it also sees one-chunk and few-element inputs)
    env = dict(os.environ, GZ_TEST_SWITCHES="small=0,scan_multi=0")
    _run_selection("big_pipeline_on_small_inputs", env,
                   "g1_cases or g3_random or g4_loader or cfg2_10k or cfg3_20k or noisy_corpus or long_and_huge or random_tables_fuzz or extreme_batch")


def _diag_library():
    """build_ab/libgz_diag.so (make diag: -DGZ_DIAG), built when it is missing or older than its sources."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "genz-tokenize_amd", "csrc"), "diag"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    return os.path.join(root, "build_ab", "libgz_diag.so")


def test_index_assertions_on_a_poisoned_workspace():
    """The diagnostic build checks every index its kernels take out of memory against the array it goes into -- miss lists, sorted
    miss records, word records and their token places, the wide / long word lists, first-word indices -- and a consumer that meets a
    record no merge kernel has finished (gz_kernels.hip GZ_CHK; a failure is recorded, the access skipped, the call fails at its
    next synchronisation).  With GZ_DIAG_POISON=1 the per-call workspace is filled with 0xFF before every call: a kernel that
    consumes a word nobody wrote in that call then reads an impossible value EVERY time.  The selection is the one of
    test_big_pipeline_on_small_inputs (round 4: one child of it aborted once, unexplained) plus the pair corpora: tiny batches, the
    load-time whole-word build with its many words of 17..32 symbols, random tables, words of up to 3 000 symbols -- all through
    the kernel pipeline, the chained scan on every size."""
    env = dict(os.environ, GZ_LIBRARY=_diag_library(), GZ_TEST_SWITCHES="small=0,scan_multi=0,diag_poison=1,brk_side=0", GZ_TABLE_CACHE="off")
    _run_selection("diag_index_assertions", env,
                   "g1_cases or g3_random or g4_loader or cfg2_10k or cfg3_20k or noisy_corpus or noisy_pairs or long_and_huge or random_tables_fuzz or extreme_batch or not_utf8")


def test_dense_hash_tables():
    """The long-key whole-word table (words of 17..32 bytes, linear probing) is built at 1/16 load, so its continue-probing
    branches almost never run on the default build.  The switch tab_slack = 2 builds it at half load: the golden batches, the 20 k-document digests, the noisy corpora (single and
    pairs), long words and the small-kernel shapes run again that way in a child process."""
    env = dict(os.environ, GZ_TEST_SWITCHES="tab_slack=2")
    _run_selection("dense_hash_tables", env,
                   "g1_cases or g3_random or cfg3_20k or noisy_corpus or noisy_pairs or long_and_huge or small_kernel_shapes or random_tables_fuzz")


def test_perfect_hash_overflow_buckets_on_the_gpu():
    """The perfectly hashed tables never need an overflow bucket on real files, so the kernels' "this bucket could not be
    placed: probe on" branches (and the merge kernel's OVF instantiation) never run on the default build.
    The switch ph_force_overflow = 3 makes the builder refuse every third bucket -- several thousand keys of both tables then sit in
    overflow buckets: the golden batches, the 20 k-document digests, the noisy corpora, long words and random tables run
    again that way in a child process (table cache off, so that the tables are really rebuilt)."""
    env = dict(os.environ, GZ_TEST_SWITCHES="ph_force_overflow=3", GZ_TABLE_CACHE="off")
    _run_selection("overflow_buckets", env,
                   "g1_cases or g3_random or cfg3_20k or noisy_corpus or noisy_pairs or long_and_huge or random_tables_fuzz")


@pytest.mark.parametrize("switches", [{"side": 0, "brk_side": 0}, {"assemble": 2}, {"assemble": 1},
                                      {"m2_split_min": 1, "m2_split_always": 1}],
                         ids=["one_stream", "lds_row_kernel", "scatter_row_kernel", "two_merge_instances"])
def test_alternative_kernels_stay_exact(switches):
    """Schedules and kernels the library keeps beside the default ones: everything on one stream (side = 0, brk_side = 0), and
    the pair-mode / ragged row writers run on dense single texts (assemble = 2: whole rows through LDS, the pair-mode kernel; 1:
    the scatter kernel of the ragged layouts); the merge kernel's two instances (words of up to 8 / up to 16 symbols, gz_hot.inc)
    sharing EVERY batch, however small and with the whole-word tables on -- by default they only do when the tables are off and
    the misses are many.  Each runs the golden batches, the 20 k-document digests, the noisy corpora, bytes that are not UTF-8
    and the long words in a child process, small batches through the pipeline."""
    env = dict(os.environ, GZ_TEST_SWITCHES=",".join(["small=0"] + ["%s=%d" % kv for kv in sorted(switches.items())]))
    _run_selection("alternative_kernels_" + "_".join("%s%s" % kv for kv in sorted(switches.items())), env,
                   "g1_cases or g3_random_batched or cfg3_20k or noisy_corpus or long_and_huge or extreme_batch or not_utf8")


def test_far_word_records_on_small_batches():
    """A merged word's record has two forms (gz_pipeline.inc, W_NEAR): "near" -- count and place of its tokens in the record
    itself, for the first 2^25 places of the compact token area -- and "far" (count in the record, place in waux).  Batches
    below some 10 M merged words only ever produce near records: the switch near_limit = 300 gives the far form to every word placed
    from 300 on, so that the golden batches, the 20 k-document digests, the noisy corpora, the word-count output and the long
    words see BOTH forms side by side, in a child process with the small batches sent through the kernel pipeline."""
    env = dict(os.environ, GZ_TEST_SWITCHES="near_limit=300,small=0")
    _run_selection("far_word_records", env,
                   "g1_cases or g3_random or cfg3_20k or noisy_corpus or noisy_pairs or long_and_huge or csr_host_path or large_noisy or extreme_batch")


def test_small_kernel_shapes(tok, oracle_tables, sampler):
    """The one-launch path on its edge shapes, against the C oracle: documents of exactly 4 096 bytes (one per workgroup),
    64 tiny documents per workgroup, empty documents, every max_len class (1, 2, 3, odd, 1 024), a long word, a word of
    more than 1 024 symbols, a batch just under the size limit of the path (2 MB); ragged layouts; and the pinned
    staging block of small host calls with pairs."""
    import gz_oracle_c as OC
    co = OC.COracle(open(corpus.VOCAB_PATH, "rb").read(), open(corpus.BPE_PATH, "rb").read())
    text, offs, _ = corpus.config_corpus(3, n_docs=3000, seed=51, sampler=sampler)
    text, offs = corpus.add_noise(text, offs, seed=8, rate=0.05)
    raw = text.tobytes()
    docs = [raw[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    docs = [d[:4096] for d in docs]
    docs = [d if len(d) < 4096 or (d[-1] & 0xC0) != 0x80 else d[:4093] for d in docs]           # keep the cut on a character boundary
    docs = [d.decode("utf-8", "ignore").encode("utf-8") for d in docs]
    big = ("kh\u00f4ng " * 700).encode("utf-8")[:4096].decode("utf-8", "ignore").encode("utf-8")
    docs += [b"", b" ", b"a", b"", big, b"x" * 4096, ("\u1ea5" * 1300).encode("utf-8"), b"zq" * 2048, b"\n".join([b"ab"] * 1300), b""]
    docs += [b"w%d" % i for i in range(300)]
    o = np.zeros(len(docs) + 1, np.int64); np.cumsum([len(d) for d in docs], out=o[1:])
    t = np.frombuffer(b"".join(docs) + b"  ", np.uint8)[:int(o[-1])]
    assert int(o[-1]) < (2 << 20)
    for ml in (1, 2, 3, 7, 64, 256, 1024):
        got = tok.encode_packed(t, o, max_len=ml)
        wi, wm, _, _, row, _, _ = co.call_packed(np.ascontiguousarray(t), o, max_len=ml)
        k = int(row[-1])
        assert np.array_equal(got["input_ids"].reshape(-1), wi[:k]), ml
        assert np.array_equal(got["attention_mask"].reshape(-1), wm[:k]), ml
        assert np.array_equal(got["n_real"], wm[:k].reshape(-1, ml).sum(1)), ml           # (no real token has the pad id here)
    # every word through the merge loop
    got = tok.encode_packed(t, o, max_len=48, word_table=False)
    wi, wm, _, _, row, _, _ = co.call_packed(np.ascontiguousarray(t), o, max_len=48)
    assert np.array_equal(got["input_ids"].reshape(-1), wi[:int(row[-1])])
    # ragged layouts: the kernel leaves unpadded rows in the raw area, finalize cuts / pads them
    for ml, pad, trunc in ((None, True, True), (64, False, True), (64, True, False), (64, False, False), (0, True, True), (-3, True, True)):
        got = tok.encode_packed(t, o, max_len=ml, padding=pad, truncation=trunc)
        wi, wm, _, _, row, _, _ = co.call_packed(np.ascontiguousarray(t), o, max_len=ml, padding=pad, truncation=trunc)
        k = int(row[-1])
        assert np.array_equal(got["row_off"], row), (ml, pad, trunc)
        assert np.array_equal(got["input_ids"], wi[:k]), (ml, pad, trunc)
        assert np.array_equal(got["attention_mask"], wm[:k]), (ml, pad, trunc)
    # pairs: the one-launch kernel with two segments per document, through the pinned staging block of small host calls
    # (<= 1 MB in + out) and through the ordinary copies (the whole batch as A, itself shifted by 7 documents as B)
    def part(lo, hi):
        return np.ascontiguousarray(t[int(o[lo]):int(o[hi])]), np.ascontiguousarray(o[lo:hi + 1] - o[lo])
    nd = len(docs)
    ta2, oa2, _ = corpus.config_corpus(2, n_docs=2000, seed=52, sampler=sampler)
    tb2, ob2, _ = corpus.config_corpus(2, n_docs=2000, seed=53, sampler=sampler)
    ta2, oa2 = corpus.add_noise(ta2, oa2, seed=9, rate=0.05)
    cases = [part(0, 400) + part(400, 800),                      # long documents: the big pipeline behind the staging block
             part(nd - 300, nd) + part(nd - 307, nd - 7),        # tiny documents, 64 pairs per workgroup, some texts empty
             (np.ascontiguousarray(ta2), np.ascontiguousarray(oa2, dtype=np.int64), np.ascontiguousarray(tb2), np.ascontiguousarray(ob2, dtype=np.int64))]
    for (ts, os_, pt, po), (ml, pad, trunc) in [(c, sh) for c in cases for sh in ((24, True, True), (None, True, True), (24, False, True),
                                                                                     (24, True, False), (5, True, True), (2, True, True))]:
        got = tok.encode_packed(ts, os_, pt, po, max_len=ml, padding=pad, truncation=trunc)
        wi, wm, wt, wq, row, plen, st = co.call_packed(ts, os_, pt, po, max_len=ml, padding=pad, truncation=trunc)
        k = int(row[-1])
        ro = np.asarray(got["row_off"], np.int64)
        assert np.array_equal(ro, row), (ml, pad, trunc)
        assert np.array_equal(np.asarray(got["input_ids"]).reshape(-1), wi[:k]), (ml, pad, trunc)
        assert np.array_equal(np.asarray(got["attention_mask"]).reshape(-1), wm[:k]), (ml, pad, trunc)
        assert np.array_equal(got["status"], st), (ml, pad, trunc)
        gl = np.asarray(got["pair_len"]).reshape(-1, 2)
        assert np.array_equal(gl, plen.reshape(-1, 2)), (ml, pad, trunc)
        assert np.array_equal(_gather(got["sequence_id"].reshape(-1), ro[:-1], gl[:, 0]), _gather(wq, row[:-1], gl[:, 0])), (ml, pad, trunc)
        assert np.array_equal(_gather(got["token_type_ids"].reshape(-1), ro[:-1], gl[:, 1]), _gather(wt, row[:-1], gl[:, 1])), (ml, pad, trunc)


def test_small_path_random_shapes(tok, sampler):
    """The one-launch path over 60 seeded random batches against the C oracle: 1..700 documents of random length classes
    (empty, a few bytes, sentences, up to 4 KiB), singles and pairs, every layout (dense, ragged with and without
    padding / truncation, max_len None / <= 0 / tiny), table on and off."""
    import random
    import gz_oracle_c as OC
    co = OC.COracle(open(corpus.VOCAB_PATH, "rb").read(), open(corpus.BPE_PATH, "rb").read())
    text, offs, _ = corpus.config_corpus(3, n_docs=1500, seed=61, sampler=sampler)
    text, offs = corpus.add_noise(text, offs, seed=11, rate=0.06)
    raw = text.tobytes().decode("utf-8")
    words = raw.split(" ")
    r = random.Random(2024)

    def doc():
        k = r.choice([0, 0, 1, 1, 2, 3, 5, 8, 13, 30, 60, 120, 400])
        i = r.randrange(len(words) - k - 1)
        return " ".join(words[i:i + k]).encode("utf-8")[:r.choice([4096, 4096, 2000, 700])].decode("utf-8", "ignore").encode("utf-8")

    def pack(docs):
        o = np.zeros(len(docs) + 1, np.int64); np.cumsum([len(d) for d in docs], out=o[1:])
        return np.frombuffer(b"".join(docs) + b" " * 16, np.uint8)[:int(o[-1])].copy(), o

    for trial in range(60):
        n = r.choice([1, 1, 2, 3, 7, 64, 65, 129, 300, 700])
        pair = r.random() < 0.5
        a = [doc() for _ in range(n)]
        b = [doc() for _ in range(n)] if pair else None
        ml, pad, trunc = r.choice([(None, True, True), (16, True, True), (33, True, True), (1, True, True), (2, True, True), (4, True, True),
                                   (256, True, True), (20, False, True), (20, True, False), (20, False, False), (0, True, True), (-2, True, True)])
        wt = r.random() < 0.8
        ta, oa = pack(a)
        if pair:
            tb, ob = pack(b)
            got = tok.encode_packed(ta, oa, tb, ob, max_len=ml, padding=pad, truncation=trunc, word_table=wt)
            wi, wm, wtt, wq, row, plen, st = co.call_packed(ta, oa, tb, ob, max_len=ml, padding=pad, truncation=trunc)
        else:
            got = tok.encode_packed(ta, oa, max_len=ml, padding=pad, truncation=trunc, word_table=wt)
            wi, wm, wtt, wq, row, plen, st = co.call_packed(ta, oa, max_len=ml, padding=pad, truncation=trunc)
        what = (trial, n, pair, ml, pad, trunc, wt)
        k = int(row[-1])
        ro = np.asarray(got["row_off"], np.int64)
        assert np.array_equal(ro, row), what
        assert np.array_equal(np.asarray(got["input_ids"]).reshape(-1)[:k], wi[:k]), what
        assert np.array_equal(np.asarray(got["attention_mask"]).reshape(-1)[:k], wm[:k]), what
        if pair:
            assert np.array_equal(got["status"], st), what
            gl = np.asarray(got["pair_len"]).reshape(-1, 2)
            assert np.array_equal(gl, plen.reshape(-1, 2)), what
            assert np.array_equal(_gather(np.asarray(got["sequence_id"]).reshape(-1), ro[:-1], gl[:, 0]), _gather(wq, row[:-1], gl[:, 0])), what
            assert np.array_equal(_gather(np.asarray(got["token_type_ids"]).reshape(-1), ro[:-1], gl[:, 1]), _gather(wtt, row[:-1], gl[:, 1])), what


def test_device_entry_points_take_absolute_offsets(tok, oracle_tables, sampler):
    """gz_preprocess_batch_device and gz_decode_batch_device read their input like gz_encode_batch_device: offsets are
    absolute from the base pointer, the first one need not be 0."""
    ctx = tok._ctx
    text, offs, L = corpus.config_corpus(2, n_docs=500, seed=41, sampler=sampler)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    lo, n = 137, 300                                                   # documents [137, 437) of the packed text
    sub = np.ascontiguousarray(offs[lo:lo + n + 1])
    assert sub[0] != 0
    nbytes = int(sub[-1] - sub[0])
    d_text = ctx.alloc(len(text) + 64); ctx.h2d(d_text, text)
    d_off = ctx.alloc(8 * (n + 1)); ctx.h2d(d_off, sub)
    d_out = ctx.alloc(nbytes + 64); d_oo = ctx.alloc(8 * (n + 1))
    kept = ctx.preprocess_device([3, 4], d_text, d_off, n, nbytes, d_out, nbytes, d_oo)        # punct, emoji
    got = np.empty(kept, np.uint8); ctx.d2h(got, d_out)
    oo = np.empty(n + 1, np.int64); ctx.d2h(oo, d_oo)
    raw = text.tobytes()
    want = [O.preprocess(raw[offs[lo + i]:offs[lo + i + 1]].decode(), ["punct", "emoji"]) for i in range(n)]
    assert [got[oo[i]:oo[i + 1]].tobytes().decode() for i in range(n)] == want
    # decode: rows [lo, lo + n) of a packed id array
    enc = tok.encode_packed(text, offs, max_len=None)
    ids, ro = enc["input_ids"], np.ascontiguousarray(enc["row_off"], dtype=np.int64)
    rsub = np.ascontiguousarray(ro[lo:lo + n + 1])
    assert rsub[0] != 0
    tok.decode_batch([[1, 2]])                                            # (takes the decoder snapshot)
    d_ids = ctx.alloc(ids.nbytes + 64); ctx.h2d(d_ids, ids)
    d_ro = ctx.alloc(8 * (n + 1)); ctx.h2d(d_ro, rsub)
    d_do = ctx.alloc(8 * (n + 1))
    unk = tok.unk_token.encode()
    need = ctx.decode_device(d_ids, d_ro, n, unk, 0, 0, d_do)
    d_txt = ctx.alloc(need + 64)
    ctx.decode_device(d_ids, d_ro, n, unk, d_txt, need, d_do)
    txt = np.empty(need, np.uint8); ctx.d2h(txt, d_txt)
    do = np.empty(n + 1, np.int64); ctx.d2h(do, d_do)
    want = [O.decode(ids[ro[lo + i]:ro[lo + i + 1]].tolist(), oracle_tables) for i in range(n)]
    assert [txt[do[i]:do[i + 1]].tobytes().decode() for i in range(n)] == want
    for p in (d_text, d_off, d_out, d_oo, d_ids, d_ro, d_do, d_txt):
        ctx.free(p)


def _oracle_rows(t, text, offs, pairs, ml, pad, tr):
    raw = text.tobytes()
    docs = [raw[offs[i]:offs[i + 1]].decode("utf-8") for i in range(len(offs) - 1)]
    pr = None
    if pairs is not None:
        praw = pairs[0].tobytes()
        pr = [praw[pairs[1][i]:pairs[1][i + 1]].decode("utf-8") for i in range(len(offs) - 1)]
    return O.call_batch(t, docs, pr, ml, pad, tr)


def _compare_batch(out, want, pair):
    I, M, T, S, st = want
    ro = out["row_off"]
    ids = out["input_ids"].reshape(-1); mask = out["attention_mask"].reshape(-1)
    for i in range(len(I)):
        a, b = int(ro[i]), int(ro[i + 1])
        assert ids[a:b].tolist() == I[i], i
        assert mask[a:b].tolist() == M[i], i
        assert int(out["status"][i]) == st[i], i
        if pair and st[i] == 0:
            ns, nt = (int(x) for x in out["pair_len"][i])
            assert out["sequence_id"].reshape(-1)[a:a + ns].tolist() == S[i], i
            assert out["token_type_ids"].reshape(-1)[a:a + nt].tolist() == T[i], i


@pytest.mark.parametrize("shape", [(128, True, True), (None, True, True), (16, True, False), (7, False, True),
                                   (1, True, True), (0, True, True), (-3, True, True)])
def test_noisy_corpus_vs_oracle(tok, oracle_tables, sampler, shape):
    ml, pad, tr = shape
    text, offs, _ = corpus.config_corpus(2, n_docs=1500, seed=77, sampler=sampler)
    text, offs = corpus.add_noise(text, offs, seed=5, rate=0.06)
    want = _oracle_rows(oracle_tables, text, offs, None, ml, pad, tr)
    for wt in (True, False):
        out = tok.encode_packed(text, offs, max_len=ml, padding=pad, truncation=tr, word_table=wt)
        _compare_batch(out, want, False)


@pytest.mark.parametrize("shape", [(64, True, True), (None, True, True), (24, True, False), (5, True, True), (2, True, True)])
def test_noisy_pairs_vs_oracle(tok, oracle_tables, sampler, shape):
    ml, pad, tr = shape
    ta, oa, _ = corpus.config_corpus(2, n_docs=800, seed=11, sampler=sampler)
    tb, ob, _ = corpus.config_corpus(2, n_docs=800, seed=12, sampler=sampler)
    ta, oa = corpus.add_noise(ta, oa, seed=1, rate=0.05)
    tb, ob = corpus.add_noise(tb, ob, seed=2, rate=0.05)
    want = _oracle_rows(oracle_tables, ta, oa, (tb, ob), ml, pad, tr)
    for wt in (True, False):
        out = tok.encode_packed(ta, oa, tb, ob, max_len=ml, padding=pad, truncation=tr, word_table=wt)
        _compare_batch(out, want, True)


def _gather(flat, off, lens):
    """Concatenate flat[off[i] : off[i] + lens[i]] for all i."""
    lens = np.asarray(lens, np.int64)
    cum = np.concatenate([[0], np.cumsum(lens)])
    return flat[np.repeat(np.asarray(off, np.int64) - cum[:-1], lens) + np.arange(cum[-1])]


@pytest.mark.parametrize("shape", [(96, True, True), (None, True, True), (40, True, False), (3, True, True)])
def test_large_noisy_vs_c_oracle(tok, sampler, shape):
    """20 k noisy documents, single and paired, against the plain-C restatement (oracle/gz_oracle.c), compared as
    whole arrays: ids, mask, row offsets, status, sequence_id and token_type_ids."""
    import gz_oracle_c as OC
    from corpus import VOCAB_PATH, BPE_PATH
    co = OC.COracle(open(VOCAB_PATH, "rb").read(), open(BPE_PATH, "rb").read())
    ml, pad, tr = shape
    ta, oa, _ = corpus.config_corpus(3, n_docs=20000, seed=31, sampler=sampler)
    tb, ob, _ = corpus.config_corpus(2, n_docs=20000, seed=32, sampler=sampler)
    ta, oa = corpus.add_noise(ta, oa, seed=3, rate=0.04)
    tb, ob = corpus.add_noise(tb, ob, seed=4, rate=0.04)
    ta, tb = np.ascontiguousarray(ta), np.ascontiguousarray(tb)
    oa, ob = np.ascontiguousarray(oa, dtype=np.int64), np.ascontiguousarray(ob, dtype=np.int64)
    for pair in (False, True):
        ids, mask, tt, sq, row, pl, st = co.call_packed(ta, oa, tb if pair else None, ob if pair else None, ml, pad, tr)
        lens = np.diff(row)
        for wt in (True, False):
            out = (tok.encode_packed(ta, oa, tb, ob, max_len=ml, padding=pad, truncation=tr, word_table=wt) if pair else
                   tok.encode_packed(ta, oa, max_len=ml, padding=pad, truncation=tr, word_table=wt))
            ro = np.asarray(out["row_off"], np.int64)
            assert np.array_equal(np.diff(ro), lens)
            assert np.array_equal(_gather(out["input_ids"].reshape(-1), ro[:-1], lens), ids[:row[-1]])
            assert np.array_equal(_gather(out["attention_mask"].reshape(-1), ro[:-1], lens), mask[:row[-1]])
            if pair:
                assert np.array_equal(np.asarray(out["status"]), st)
                assert st.sum() > 0 or ml is None or ml > 8
                gl = np.asarray(out["pair_len"]).reshape(-1, 2)
                assert np.array_equal(gl, pl.reshape(-1, 2))
                assert np.array_equal(_gather(out["sequence_id"].reshape(-1), ro[:-1], gl[:, 0]), _gather(sq, row[:-1], gl[:, 0]))
                assert np.array_equal(_gather(out["token_type_ids"].reshape(-1), ro[:-1], gl[:, 1]), _gather(tt, row[:-1], gl[:, 1]))


def test_long_and_huge_words(tok, oracle_tables):
    """Words of 17..1024 symbols take the wave-cooperative LDS path, longer ones the global arena pass; a word may
    also straddle or fill whole 1-KiB tiles."""
    import random
    r = random.Random(9)
    alpha = "nghiêngtrườngkhôngaaaabđ_.😀"
    docs = []
    for n in [17, 18, 31, 63, 64, 65, 100, 255, 256, 257, 700, 1023, 1024, 1025, 1100, 2047, 2048, 2500, 3000]:
        w = "".join(r.choice(alpha) for _ in range(n))
        docs.append("xin chào " + w + " việt nam\n" + w[: n // 2] + "\nhết")
        docs.append(w)
        docs.append(" " * (1000 - n % 7) + w + "\n")
    docs.append("a" * 1500 + " " + "ng" * 600 + "\n" + "h" * 1024)
    docs.append("")
    for ml, pad, tr in [(None, True, True), (64, True, True), (4096, True, True)]:
        out = tok.encode_batch(docs, max_len=ml, padding=pad, truncation=tr)
        want = O.call_batch(oracle_tables, docs, None, ml, pad, tr)
        _compare_batch(out, want, False)
    pairs = list(reversed(docs))
    out = tok.encode_batch(docs, pairs, max_len=None)
    _compare_batch(out, O.call_batch(oracle_tables, docs, pairs, None, True, True), True)


def test_bytes_that_are_not_utf8_stay_inside_their_document(tok, sampler):
    """Input outside the contract (the C ABI takes raw bytes): documents full of stray continuation bytes, truncated and
    over-long sequences sit BETWEEN well-formed documents.  What they tokenize to is not specified -- but their words must
    not spill into a neighbour's tokens (the merge kernel reserves token places from a lead-byte count; a code point that
    starts at a continuation byte takes the wave-cooperative path instead), the call must not fault, and two runs must agree.
    The well-formed documents must come out exactly as in a batch without the bad ones, whole-word tables on and off, through
    the big pipeline and the one-launch kernel."""
    import random
    text, offs, _ = corpus.config_corpus(3, n_docs=3000, seed=91, sampler=sampler)
    raw = text.tobytes()
    r = random.Random(17)
    def junk():
        n = r.choice([1, 2, 3, 5, 8, 13, 16, 17, 31, 40, 70, 200])
        pool = [b"\x80", b"\xbf", b"\x9a", b"a", b"\xe1", b"\xe1\xba", b"\xf0\x9f", b"\xc3", b"\xc2\xa0", b" ", b"\n", b"ng", b"\xe1\xbb\x87", b"\xff", b"\xc0\x80"]
        return b"".join(r.choice(pool) for _ in range(n))
    docs, good = [], []
    for i in range(len(offs) - 1):
        if r.random() < 0.3:
            docs.append(junk())
        good.append(len(docs))
        docs.append(raw[offs[i]:offs[i + 1]])
    docs.append(junk())
    mixed = np.frombuffer(b"".join(docs), dtype=np.uint8)
    moffs = np.concatenate([[0], np.cumsum([len(d) for d in docs])]).astype(np.int64)
    for wt in (True, False):
        for ml in (48, 256):
            want = tok.encode_packed(text, offs, max_len=ml, word_table=wt)
            a = tok.encode_packed(mixed, moffs, max_len=ml, word_table=wt)
            b = tok.encode_packed(mixed, moffs, max_len=ml, word_table=wt)
            assert np.array_equal(a["input_ids"], b["input_ids"]) and np.array_equal(a["attention_mask"], b["attention_mask"])
            assert np.array_equal(a["input_ids"][good], want["input_ids"]) and np.array_equal(a["attention_mask"][good], want["attention_mask"])
    # small batches take the one-launch kernel: the same property on the first documents
    k = 40
    sm = tok.encode_packed(mixed[:moffs[k]], moffs[:k + 1], max_len=32)
    gk = [g for g in good if g < k]
    ref = tok.encode_packed(text[:offs[len(gk)]], offs[:len(gk) + 1], max_len=32)
    assert np.array_equal(sm["input_ids"][gk], ref["input_ids"])


def test_deterministic(tok, sampler):
    text, offs, L = corpus.config_corpus(3, n_docs=20000, seed=21, sampler=sampler)
    a = tok.encode_packed(text, offs, max_len=L)
    b = tok.encode_packed(text, offs, max_len=L)
    assert np.array_equal(a["input_ids"], b["input_ids"]) and np.array_equal(a["attention_mask"], b["attention_mask"])


def test_rccl_gather_rows_single_rank(tok):
    """The multi-GPU exchange step of the C ABI with a one-rank communicator: RCCL loads, the communicator
    initialises, the root's own block lands in the gathered buffer (more ranks cannot run on a one-GPU box)."""
    ctx = tok._ctx
    uid = ctx.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_init(uid, 0, 1)
    rows = np.arange(37 * 16, dtype=np.int32).reshape(37, 16)
    d_src = ctx.alloc(rows.nbytes); d_dst = ctx.alloc(rows.nbytes)
    ctx.h2d(d_src, rows)
    ctx.gather_rows(d_src, 37, 16, d_dst, [37], 0)
    ctx.sync()
    back = np.zeros_like(rows)
    ctx.d2h(back, d_dst)
    assert np.array_equal(back, rows)
    ctx.free(d_src); ctx.free(d_dst)


def _gpu_count():
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


def test_rccl_gather_two_ranks(tmp_path):
    """gz_gather_rows with world = 2: two fresh processes, one GPU each, uneven row counts, the exchange of call 0
    overlapped with call 1 (gz_exchange_select(1)), root != peer offsets checked against the oracle at the root, and a
    mismatched count refused before any RCCL group opens.  Needs two GPUs (RCCL refuses two ranks on one device:
    "Duplicate GPU detected"); the driver's 8-GPU bench run covers the same path through bench.py's per-peer digests."""
    import subprocess
    import sys
    if _gpu_count() < 2:
        pytest.skip("needs 2 GPUs; RCCL refuses two ranks on one device")
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gather_child.py")
    procs = [subprocess.Popen([sys.executable, child, str(r), "2", str(tmp_path)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    for r in range(2):
        assert (tmp_path / ("verdict_%d" % r)).read_text() == "ok"


def test_table_cache_hit_is_identical_and_bad_files_are_refused(tmp_path):
    """gz_load_tables behind the table cache: a second load of the same files is a HIT whose device tables are byte-identical
    to the built ones (gz_table_digest) and which tokenizes like them; a corrupted file, a truncated file and a file of
    another layout version are refused (rebuilt, rewritten), and so is a forged file whose key, length and checksum fit but
    whose content points outside the tables; other files get another key; a shared or impossible cache directory is not used."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r"""
import glob, os, sys, time
root = %r
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "genz-tokenize_amd"))
import numpy as np
import corpus
from genz_tokenize import Tokenize
cdir = os.environ["GZ_TABLE_CACHE"]
def load(**kw):
    t0 = time.perf_counter()
    t = Tokenize(**kw); t._sync_tables()
    return t, time.perf_counter() - t0
a, ta = load()
assert a._ctx.table_cache_status() == 2, a._ctx.table_cache_status()            # miss: built, written
files = glob.glob(os.path.join(cdir, "*.gztab"))
assert len(files) == 1
b, tb = load()
assert b._ctx.table_cache_status() == 1                                          # hit
assert a._ctx.table_digest() == b._ctx.table_digest()
assert a.encoder == b.encoder and a.bpe_ranks == b.bpe_ranks
text, offs, L = corpus.config_corpus(3, n_docs=3000, seed=21)
ra, rb = a.encode_packed(text, offs, max_len=64), b.encode_packed(text, offs, max_len=64)
assert np.array_equal(ra["input_ids"], rb["input_ids"]) and np.array_equal(ra["attention_mask"], rb["attention_mask"])
assert b("sinh_viên công_nghệ", "hello", max_len=10)["input_ids"] == [1, 770, 1444, 2, 2, 30469, 2, 0, 0, 0]
good = open(files[0], "rb").read()
def damaged(blob):
    open(files[0], "wb").write(blob)
    t, _ = load()
    st = t._ctx.table_cache_status()
    assert t._ctx.table_digest() == a._ctx.table_digest()
    assert open(files[0], "rb").read() == good                                   # rewritten from the rebuilt tables
    return st
flip = bytearray(good); flip[len(flip) // 2] ^= 0x40
assert damaged(bytes(flip)) == 3                                                 # one flipped bit in the payload
assert damaged(good[:len(good) - 100]) == 3                                      # truncated
assert damaged(good + b"x") == 3                                                 # trailing bytes
stale = bytearray(good); stale[4] ^= 0x01
assert damaged(bytes(stale)) == 3                                                # another layout version
# a FORGED file -- key, length and checksum all fit, but a whole-word entry claims a 31-byte key in a 16-byte slot: every
# index-bearing field is range-checked on load (the last tabp entry sits before the perfect-hash description and n_words)
import struct
M64 = (1 << 64) - 1
def sum64(p):
    a, b, n, i = 0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F, len(p), 0
    while i + 16 <= n:
        x, y = struct.unpack_from("<QQ", p, i)
        a = ((a ^ x) * 0xFF51AFD7ED558CCD) & M64; a = ((a << 27) | (a >> 37)) & M64
        b = ((b ^ y) * 0xC4CEB9FE1A85EC53) & M64; b = ((b << 31) | (b >> 33)) & M64
        i += 16
    while i < n:
        a = ((a ^ p[i]) * 0x100000001B3) & M64; i += 1
    a ^= (b + n) & M64
    a ^= a >> 33; a = (a * 0xFF51AFD7ED558CCD) & M64; a ^= a >> 33
    return a
HDR = 56                                                                         # magic, layout, key[32], payload_len, payload_sum
payload = bytearray(good[HDR:])
assert sum64(bytes(payload)) == struct.unpack_from("<Q", good, 48)[0]             # (this Python restatement of the checksum is right)
tail = 8 + (8 + 1024 * 16)                                                        # n_words, the LDS hot words (count + 1 024 entries of 16 bytes)
assert struct.unpack_from("<Q", payload, len(payload) - tail)[0] == 1024
nb = struct.unpack_from("<I", payload, len(payload) - tail - 28)[0]               # nbuckets of the word table's perfect hash
meta_at = len(payload) - tail - 28 - (8 + 2 * nb) - 32 + 12
meta = struct.unpack_from("<I", payload, meta_at)[0]
assert 1 <= (meta & 31) <= 16
struct.pack_into("<I", payload, meta_at, (meta & ~31) | 31)
forged = bytearray(good[:HDR]) + payload
struct.pack_into("<Q", forged, 48, sum64(bytes(payload)))
assert damaged(bytes(forged)) == 3                                               # consistent, but out of range: refused
c2, _ = load(unk_token="<unknown>")                                              # other specials: another key, another file
assert c2._ctx.table_cache_status() == 2 and len(glob.glob(os.path.join(cdir, "*.gztab"))) == 2
# a directory other users can write to is not used, and a directory that cannot be made is reported (status 4: rebuilt, not written)
shared = os.path.join(os.path.dirname(cdir), "shared"); os.mkdir(shared); os.chmod(shared, 0o777)
os.environ["GZ_TABLE_CACHE"] = shared
d1, _ = load()
assert d1._ctx.table_cache_status() == 4 and not os.listdir(shared) and d1._ctx.table_digest() == a._ctx.table_digest()
os.environ["GZ_TABLE_CACHE"] = os.path.join(files[0], "below_a_file")
d2, _ = load()
assert d2._ctx.table_cache_status() == 4 and d2._ctx.table_digest() == a._ctx.table_digest()
os.environ["GZ_TABLE_CACHE"] = os.path.join(os.path.dirname(cdir), "deep", "er", "cache")      # parents are created (0700)
d3, _ = load()
assert d3._ctx.table_cache_status() == 2 and (os.stat(os.environ["GZ_TABLE_CACHE"]).st_mode & 0o777) == 0o700
# a symbolic link to a good cache directory is not followed (the directory must be a real one of this user)
link = os.path.join(os.path.dirname(cdir), "link"); os.symlink(cdir, link)
os.environ["GZ_TABLE_CACHE"] = link
d4, _ = load()
assert d4._ctx.table_cache_status() == 4 and d4._ctx.table_digest() == a._ctx.table_digest()
print("ok build %%.3f s, cached %%.3f s" %% (ta, tb))
""" % root
    env = dict(os.environ, GZ_TABLE_CACHE=str(tmp_path / "cache"))
    r = _run_child([sys.executable, "-c", child], "table_cache", env=env)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("ok"), r.report


def test_encode_batch_large_path_equals_dense_path(tok):
    """Tokenize.encode_batch on 25 000 str (threaded packing -> CSR host path in a pinned arena -> dense arrays filled by host
    threads) against encode_packed (dense device path) on the same documents, twice (the arena is reused), with and without
    the whole-word tables."""
    import corpus
    text, offs, _ = corpus.config_corpus(3, n_docs=25000, seed=4)
    raw = text.tobytes()
    docs = [raw[offs[i]:offs[i + 1]].decode("utf-8") for i in range(len(offs) - 1)]
    want = tok.encode_packed(text, offs, max_len=48)
    for wt in (True, False, True):
        got = tok.encode_batch(docs, max_len=48, word_table=wt)
        assert got["input_ids"].shape == (25000, 48) and got["input_ids"].dtype == np.int32
        assert np.array_equal(got["input_ids"], want["input_ids"]) and np.array_equal(got["attention_mask"], want["attention_mask"])
        assert np.array_equal(got["n_real"], want["n_real"])
    docs[17] = 3
    with pytest.raises(TypeError, match="expected string or bytes-like object"):
        tok.encode_batch(docs, max_len=48)


def test_batch_return_offset_equals_single_calls(tok, sampler):
    """`encode_batch(..., return_offset=True)`: every document's 'offset' list (one span per word from the device's per-word
    piece counts; pair mode: B's entries shifted by A's ENTRY count, tokenize.py:231-234) equals the list `__call__` returns for
    that document alone -- the form the reference-generated G1 / G3 rows pin.  Noisy documents, empty and whitespace-only
    ones, glued line feeds, long words; single texts and pairs; dense and ragged layouts."""
    text, offs, _ = corpus.config_corpus(3, n_docs=160, seed=77, sampler=sampler)
    text, offs = corpus.add_noise(text, offs, seed=5, rate=0.05)
    raw = text.tobytes()
    docs = [raw[offs[i]:offs[i + 1]].decode("utf-8", "ignore") for i in range(len(offs) - 1)]
    docs += ["", " ", "\n", "a", "ab\ncd\n\nef", "x" * 70 + " y", "\u1ea5" * 40, "kh\u00f4ng \u0111\u01b0\u1ee3c", "a\u2003b\u00a0c"]
    pairs = docs[7:] + docs[:7]
    for ml, pad, trunc in ((32, True, True), (None, True, True), (16, False, True)):
        r = tok.encode_batch(docs, max_len=ml, padding=pad, truncation=trunc, return_offset=True)
        assert len(r["offset_off"]) == len(docs) + 1 and r["offset"].shape == (int(r["offset_off"][-1]), 2)
        for i, d in enumerate(docs):
            assert tok.offsets_of(r, i) == tok(d, max_len=ml, padding=pad, truncation=trunc, return_offset=True)["offset"], (ml, i, d[:40])
    r = tok.encode_batch(docs, pairs, max_len=None, return_offset=True)
    for i, (a, b) in enumerate(zip(docs, pairs)):
        assert tok.offsets_of(r, i) == tok(a, b, return_offset=True)["offset"], (i, a[:30], b[:30])
    # without the switch the result has no such keys, and the large-batch path is not taken with it
    assert "offset" not in tok.encode_batch(docs[:4], max_len=8)


def test_pinned_array_outlives_its_context():
    """A pinned array (gz_host_alloc) dropped AFTER its context was closed: gz_host_free must not touch the freed context."""
    import gc
    from genz_tokenize import Tokenize
    t2 = Tokenize()
    t2._sync_tables()
    arr = t2._ctx.pinned_empty(1 << 20, np.uint8)
    arr[:] = 7
    t2._ctx.close()
    assert int(arr[12345]) == 7                       # still mapped
    del arr
    gc.collect()                                       # the finalizer runs now, on a context that no longer exists


def test_null_text_with_offsets_is_an_error_not_a_crash(tok):
    import ctypes as C
    from genz_tokenize import _native
    tok._sync_tables()
    ctx = tok._ctx
    off = np.array([0, 5, 9], dtype=np.int64)
    ids = np.zeros(64, dtype=np.int32); mask = np.zeros(64, dtype=np.int32)
    rc = ctx.lib.gz_encode_batch(ctx.handle, None, off.ctypes.data_as(C.c_void_p), None, None, 2, 8,
                                 _native.GZ_PADDING | _native.GZ_TRUNCATION, 64, ids.ctypes.data_as(C.c_void_p),
                                 mask.ctypes.data_as(C.c_void_p), None, None, None, None, None, None)
    assert rc == -1 and b"NULL" in ctx.lib.gz_last_error(ctx.handle)


def test_scan_time_out_surfaces_from_every_host_path(tmp_path):
    """A look-back time-out of the chained scan (forced in the diagnostic build: gz_diag_set(1, -1)) must come back as an
    error from the CSR path and from a CHAIN of dense device calls -- also when the call that timed out is not the last
    one of the chain."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _diag_library()
    child = r"""
import os, sys, ctypes as C
import numpy as np
root = %r
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "genz-tokenize_amd"))
import corpus
import gz_switches
from genz_tokenize import Tokenize, _native
assert gz_switches.apply() == [("scan_multi", 0)]
tok = Tokenize(); tok._sync_tables(); ctx = tok._ctx
text, offs, L = corpus.config_corpus(3, n_docs=60000)           # (scan_multi = 0: every block-count scan is the chained kernel)
offs = np.ascontiguousarray(offs, dtype=np.int64)
toks, nr = ctx.encode_csr(text, offs, L, 16)                    # healthy first
n_tok = len(toks)
assert ctx.lib.gz_diag_set(1, -1) == 0                          # every look-back "times out" from now on
try:
    ctx.encode_csr(text, offs, L, 16)
    print("FAIL: no error from the CSR path"); sys.exit(1)
except _native.GzError as e:
    assert "timed out" in str(e), str(e)
# a chain of dense device calls: the FIRST one times out, the second runs healthy; the chain's close must still report it
n = len(offs) - 1
d_text = ctx.alloc(len(text) + 64); ctx.h2d(d_text, text)
d_off = ctx.alloc(8 * (n + 1)); ctx.h2d(d_off, offs)
d_ids = ctx.alloc(4 * n * L); d_mask = ctx.alloc(4 * n * L); d_nr = ctx.alloc(4 * n)
flags = _native.GZ_PADDING | _native.GZ_TRUNCATION
ctx.encode_device(d_text, d_off, 0, 0, n, L, flags, n * L, d_ids, d_mask, d_n_real=d_nr, h_text_off=offs)
assert ctx.lib.gz_diag_set(1, 1 << 21) == 0
ctx.encode_device(d_text, d_off, 0, 0, n, L, flags, n * L, d_ids, d_mask, d_n_real=d_nr, h_text_off=offs)
try:
    ctx.sync()
    print("FAIL: the chain's time-out was lost"); sys.exit(1)
except _native.GzError as e:
    assert "timed out" in str(e), str(e)
toks2, _ = ctx.encode_csr(text, offs, L, 16)                    # and the context works again afterwards
assert len(toks2) == n_tok and np.array_equal(toks2, toks)
print("ok")
""" % root
    env = dict(os.environ, GZ_LIBRARY=os.path.join(root, "build_ab", "libgz_diag.so"), GZ_TEST_SWITCHES="scan_multi=0")
    r = _run_child([sys.executable, "-c", child], "scan_time_out", env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.report


@pytest.mark.parametrize("world", [2, 4])
def test_bench_multirank_exchange_on_one_gpu_over_gloo(world):
    """The N > 1 code of bench.py -- shard ownership, gz_compact_block, exchange_select, per-round block sizes / offsets over the
    gloo control group, per-peer expand + verification at the root -- executed end to end with `world` ranks on ONE GPU:
    the transport is gloo (D2H -> send/recv -> H2D) because RCCL refuses two ranks on one device; everything else is the
    product's.  bench.py exits non-zero when any rank's shard or any gathered block differs from the C oracle."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--docs", "160000", "--transport", "gloo", "--device", "0"]
    r = _run_child(cmd, "bench_gloo_world%d" % world, env=env, timeout=420, cwd=root)
    assert r.returncode == 0, r.report
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == world and out["config"]["shards_per_rank"] == 8 // world
    assert "gloo" in out["config"]["sharding"]
    x = out["exchange"]                                          # the exchange step, taken apart (max over ranks)
    assert x["bytes_per_peer"] > 0 and x["compact_sync_ms"] > 0 and x["size_exchange_ms"] > 0 and x["gather_enqueue_ms"] > 0
    assert x["gather_enqueue_to_done_ms"] is None and isinstance(x["hidden_under_kernels"], bool)      # (device timing: RCCL transport only)
    assert len([c for c in out["verified_items"] if c.startswith("gathered shard")]) == 8     # every shard arrived and was checked
    assert out["value"] > 0


def test_bench_exchange_step_over_rccl_with_one_rank():
    """`bench.py --force-exchange` under torch.distributed.run with ONE rank and the product transport: the library's own RCCL
    communicator (ncclCommInitRank, gz_gather_rows on the exchange stream, double-buffered gz_exchange_select) beside the gloo
    control group, exactly the code path of N > 1 -- all this box can run of it (RCCL refuses two ranks on one device).  Every
    gathered block is expanded and verified against the C oracle; bench.py exits non-zero on any difference."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--docs", "160000", "--force-exchange", "--no-secondary"]
    r = _run_child(cmd, "bench_rccl_one_rank", env=env, timeout=420, cwd=root)
    assert r.returncode == 0, r.report
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and "RCCL gatherv" in out["config"]["sharding"]
    x = out["exchange"]                                          # ... with the gathers timed on the exchange stream (gz_exchange_timing_history)
    assert x["bytes_per_peer"] > 0 and x["gather_enqueue_to_done_ms"] > 0 and x["gather_enqueue_to_done_ms_max"] >= x["gather_enqueue_to_done_ms"]
    assert x["compact_sync_ms"] > 0 and x["size_exchange_ms"] > 0 and isinstance(x["hidden_under_kernels"], bool)
    assert len([c for c in out["verified_items"] if c.startswith("gathered shard")]) == 8
    assert out["value"] > 0


def test_compact_block_round_trip(tok):
    """gz_compact_block / gz_expand_block (one rank's message of the exchange step: [row lengths | real entries]) against the
    dense rows they came from, 16- and 32-bit entries."""
    import corpus
    ctx = tok._ctx
    text, offs, _ = corpus.config_corpus(3, n_docs=3000, seed=77)
    L = 64
    r = tok.encode_packed(text, offs, max_len=L)
    ids, nr = np.ascontiguousarray(r["input_ids"]), np.ascontiguousarray(r["n_real"] if "n_real" in r else r["attention_mask"].sum(axis=1).astype(np.int32))
    n = ids.shape[0]
    d_ids, d_nr = ctx.alloc(ids.nbytes), ctx.alloc(4 * n)
    ctx.h2d(d_ids, ids); ctx.h2d(d_nr, nr.astype(np.int32))
    for bits in (16, 32):
        d_blk = ctx.alloc(4 * (2 * n + n * L)); d_i2, d_m2 = ctx.alloc(ids.nbytes), ctx.alloc(ids.nbytes)
        total = ctx.compact_block(d_ids, d_nr, n, L, d_blk, bits=bits)
        assert total == int(nr.sum())
        head = np.empty(n, dtype=np.int32); ctx.sync(); ctx.d2h(head, d_blk)
        assert np.array_equal(head, nr)
        ctx.expand_block(d_blk, n, L, d_i2, d_m2, bits=bits, total=total); ctx.sync()
        i2 = np.empty_like(ids); m2 = np.empty_like(ids); ctx.d2h(i2, d_i2); ctx.d2h(m2, d_m2)
        assert np.array_equal(i2, ids) and np.array_equal(m2, r["attention_mask"])
        # a block announced with FEWER entries than its rows claim (truncated on the way, or written in another layout): the rows that
        # do not fit are written as padding, nothing beyond the announced entries is read, and the next synchronisation says so
        if total > 10:
            ctx.expand_block(d_blk, n, L, d_i2, d_m2, bits=bits, total=total // 2)
            with pytest.raises(__import__("genz_tokenize")._native.GzError) as ei:
                ctx.sync()
            assert "exchange block" in str(ei.value)
            ctx.d2h(i2, d_i2)
            first = np.zeros(n + 1, dtype=np.int64); np.cumsum(nr, out=first[1:])
            fits = first[1:] <= total // 2
            assert fits.any() and not fits.all()
            assert np.array_equal(i2[fits], ids[fits]) and (i2[~fits] == tok._special_ids()[0]).all()
            ctx.sync()                                                 # (reported once, then cleared)
        for q in (d_blk, d_i2, d_m2):
            ctx.free(q)
    ctx.free(d_ids); ctx.free(d_nr)


@pytest.mark.parametrize("case", ["rows1", "rows1_long", "pairs", "small", "odd_len"])
def test_encode_emits_its_exchange_block(tok, sampler, case):
    """gz_encode_emit_block: the encode call leaves its exchange block [n_real | first | entries] itself (the scan of the row lengths and
    the compact kernel behind its own kernels, on its own stream) -- single texts with max_len 64 (the scan over many workgroups) and
    256, rows cut at max_len among them, pairs, a batch the one-launch kernel takes, a row length that is no multiple of four.  The
    block expanded on the "receiving side" must be the call's dense output; gz_block_total must be the sum of n_real; 16- and 32-bit
    entries; chained calls keep their blocks apart (back = 0, 1, 2)."""
    from genz_tokenize import _native
    ctx = tok._ctx
    n, L = {"rows1": (40000, 64), "rows1_long": (9000, 256), "pairs": (4000, 48), "small": (40, 32), "odd_len": (5000, 30)}[case]
    ta, oa, _ = corpus.config_corpus(3, n_docs=n, seed=91, sampler=sampler)
    oa = np.ascontiguousarray(oa, dtype=np.int64); ta = np.ascontiguousarray(ta)
    pair = case == "pairs"
    want = tok.encode_packed(ta, oa, ta, oa, max_len=L) if pair else tok.encode_packed(ta, oa, max_len=L)
    ids, mask, nr = want["input_ids"].reshape(n, L), want["attention_mask"].reshape(n, L), np.asarray(want["n_real"])
    assert case != "rows1" or (nr == L).any()                   # (some rows are cut)
    d_t = ctx.alloc(len(ta) + 64); ctx.h2d(d_t, ta)
    d_o = ctx.alloc(8 * (n + 1)); ctx.h2d(d_o, oa)
    outs = [[ctx.alloc(4 * n * L) for _ in range(4)] + [ctx.alloc(8 * n + 8), ctx.alloc(4 * n + 4), ctx.alloc(4 * n + 4)] for _ in range(3)]
    flags = _native.GZ_PADDING | _native.GZ_TRUNCATION
    for bits in (16, 32):
        blocks = [ctx.alloc(4 * (2 * n + n * L) + 64) for _ in range(3)]
        for k in range(3):                                         # three chained calls, each with its own block
            d_i, d_m, d_tt, d_sq, d_pl, d_r, d_st = outs[k]
            ctx.encode_emit_block(blocks[k], bits)
            kw = dict(d_tt=d_tt, d_seq=d_sq, d_pair_len=d_pl, d_status=d_st, h_pair_off=oa) if pair else {}
            ctx.encode_device(d_t, d_o, d_t if pair else 0, d_o if pair else 0, n, L, flags, n * L, d_i, d_m, d_n_real=d_r, h_text_off=oa, **kw)
        for k in range(3):
            assert ctx.block_total(2 - k) == int(nr.sum())
            ctx.exchange_select(2 - k)
            d_i2, d_m2 = ctx.alloc(4 * n * L), ctx.alloc(4 * n * L)
            ctx.expand_block(blocks[k], n, L, d_i2, d_m2, bits=bits, total=int(nr.sum())); ctx.sync()
            i2 = np.empty((n, L), np.int32); m2 = np.empty((n, L), np.int32); head = np.empty(n, np.int32)
            ctx.d2h(i2, d_i2); ctx.d2h(m2, d_m2); ctx.d2h(head, blocks[k])
            assert np.array_equal(head, nr) and np.array_equal(i2, ids) and np.array_equal(m2, mask)
            ctx.free(d_i2); ctx.free(d_m2)
        ctx.exchange_select(0)
        for b in blocks:
            ctx.free(b)
    # a call without the arming has no block; a ragged call refuses the arming
    ctx.encode_device(d_t, d_o, 0, 0, n, L, flags, n * L, outs[0][0], outs[0][1], d_n_real=outs[0][5], h_text_off=oa); ctx.sync()
    with pytest.raises(_native.GzError):
        ctx.block_total(0)
    blk = ctx.alloc(4 * (2 * n + n * L) + 64)
    ctx.encode_emit_block(blk, 32)
    with pytest.raises(_native.GzError):
        ctx.encode_device(d_t, d_o, 0, 0, n, 0, _native.GZ_MAX_LEN_NONE, len(ta) + 2 * n, outs[0][0], outs[0][1], d_row_off=outs[0][4], d_n_real=outs[0][5], h_text_off=oa)
    ctx.sync()                                                       # (the refused call used the arming up: the next call emits nothing)
    ctx.encode_device(d_t, d_o, 0, 0, n, L, flags, n * L, outs[0][0], outs[0][1], d_n_real=outs[0][5], h_text_off=oa); ctx.sync()
    with pytest.raises(_native.GzError):
        ctx.block_total(0)
    # the arming belongs to the next DEVICE call only: host calls in between (dense and ragged) and a table reload neither consume
    # it nor fail on it, nor write into the block (sized for another call)
    if case == "rows1":
        ctx.encode_emit_block(blk, 32)
        few = 50
        h = tok.encode_packed(ta[:oa[few]], oa[:few + 1], max_len=L)                       # dense host call
        assert np.array_equal(h["input_ids"], ids[:few])
        h = tok.encode_packed(ta[:oa[few]], oa[:few + 1])                                  # ragged host call: no "needs a dense call"
        assert int(h["row_off"][-1]) >= few * 2
        data = os.path.join(os.path.dirname(_native.__file__), "data")
        ctx.load_tables(open(os.path.join(data, "vocab.txt"), "rb").read(), open(os.path.join(data, "bpe.codes"), "rb").read(),
                        [tok.pad_token, tok.bos_token, tok.eos_token, tok.mask_token, tok.unk_token])   # (encodes every candidate word through the same function)
        ctx.encode_device(d_t, d_o, 0, 0, n, L, flags, n * L, outs[0][0], outs[0][1], d_n_real=outs[0][5], h_text_off=oa)
        assert ctx.block_total(0) == int(nr.sum())                  # the device call still had its arming
        d_i2, d_m2 = ctx.alloc(4 * n * L), ctx.alloc(4 * n * L)
        ctx.expand_block(blk, n, L, d_i2, d_m2, bits=32, total=int(nr.sum())); ctx.sync()
        i2 = np.empty((n, L), np.int32); ctx.d2h(i2, d_i2)
        assert np.array_equal(i2, ids)
        ctx.free(d_i2); ctx.free(d_m2)
    ctx.free(blk)
    for q in [d_t, d_o] + [x for o in outs for x in o]:
        ctx.free(q)


def test_allocation_failure_inside_an_entry_point_is_an_error_code(tok, sampler):
    """No C++ exception crosses the C ABI: every extern "C" body is a function-try-block.  The switch inject_bad_alloc makes the
    k-th allocation site (where an entry point grows a std::vector) throw std::bad_alloc: the call must answer GZ_E_NOMEM -- not
    std::terminate, which is a silent SIGABRT -- and the context must work afterwards."""
    from genz_tokenize import _native
    ctx = tok._ctx
    n, L = 30000, 32
    ta, oa, _ = corpus.config_corpus(3, n_docs=n, seed=5, sampler=sampler)
    ta = np.ascontiguousarray(ta); oa = np.ascontiguousarray(oa, dtype=np.int64)
    want = tok.encode_packed(ta, oa, max_len=L)
    calls = {
        "dense host call (sub-batches, rows padded on host threads)": lambda: tok.encode_packed(ta, oa, max_len=L),
        "dense host call, rows over the bus": lambda: tok.encode_packed(ta, oa, ta, oa, max_len=L),
        "csr host call": lambda: tok.encode_packed_csr(ta, oa, max_len=L),
        "word counts": lambda: tok.encode_batch(["a b c", "d"] * 300, return_offset=True),
        "table digest": lambda: ctx.table_digest(),
    }
    for what, call in calls.items():
        hit = 0
        for k in range(1, 8):
            _native.debug_set("inject_bad_alloc", k, ctx)
            try:
                call()
            except _native.GzError as e:
                assert e.code == _native.GZ_E_NOMEM, (what, k, str(e))
                hit += 1
            finally:
                _native.debug_set("inject_bad_alloc", 0, ctx)
        assert hit >= 1, what                                         # (at least the first site of every call is reached)
        r = tok.encode_packed(ta, oa, max_len=L)                      # ... and the context is whole afterwards
        assert np.array_equal(r["input_ids"], want["input_ids"]) and np.array_equal(r["n_real"], want["n_real"])


def test_gather_rows_refuses_bad_arguments_before_opening_a_group(tok):
    """Count / pointer mistakes come back as GZ_E_INVALID without touching RCCL (a rank that fails inside an open group
    hangs the others); the communicator keeps working afterwards."""
    from genz_tokenize import _native
    ctx = tok._ctx
    ctx.comm_init(ctx.comm_unique_id(), 0, 1)
    rows = np.arange(5 * 8, dtype=np.int32).reshape(5, 8)
    d_src = ctx.alloc(rows.nbytes); d_dst = ctx.alloc(rows.nbytes)
    ctx.h2d(d_src, rows)
    for args in ((d_src, 4, 8, d_dst, [5], 0),        # local count != rows_per_rank[rank]
                 (d_src, 5, 8, 0, [5], 0),            # root without a receive buffer
                 (0, 5, 8, d_dst, [5], 0),            # rows to send, no send buffer
                 (d_src, 5, 8, d_dst, [5], 1)):       # root out of range
        with pytest.raises(_native.GzError):
            ctx.gather_rows(*args)
    ctx.gather_rows(d_src, 5, 8, d_dst, [5], 0)
    ctx.sync()
    back = np.zeros_like(rows); ctx.d2h(back, d_dst)
    assert np.array_equal(back, rows)
    ctx.free(d_src); ctx.free(d_dst)


@pytest.mark.parametrize("n_docs", [5000, 40003])
def test_compact_expand_round_trip(tok, sampler, n_docs):
    """The exchange step's compact form: rows without padding -> padding and mask rebuilt == the dense output.  (40 003 rows: the row
    offsets come from the scan over many workgroups, as for a shard; 5 000: from the one-workgroup scan.)"""
    ctx = tok._ctx
    text, offs, L = corpus.config_corpus(3, n_docs=n_docs, seed=31, sampler=sampler)
    out = tok.encode_packed(text, offs, max_len=L)
    ids, mask, n_real = out["input_ids"], out["attention_mask"], out["n_real"]
    n = len(offs) - 1
    d_ids = ctx.alloc(ids.nbytes); d_nr = ctx.alloc(n_real.nbytes); d_comp = ctx.alloc(ids.nbytes)
    d_i2 = ctx.alloc(ids.nbytes); d_m2 = ctx.alloc(ids.nbytes)
    ctx.h2d(d_ids, ids); ctx.h2d(d_nr, n_real)
    total = ctx.compact_rows(d_ids, d_nr, n, L, d_comp)
    assert total == int(n_real.sum())
    comp = np.empty(total, dtype=np.int32); ctx.d2h(comp, d_comp)
    assert np.array_equal(comp, np.concatenate([ids[i, :n_real[i]] for i in range(n)]))
    ctx.expand_rows(d_comp, d_nr, n, L, d_i2, d_m2)
    ctx.sync()
    i2 = np.empty_like(ids); m2 = np.empty_like(mask)
    ctx.d2h(i2, d_i2); ctx.d2h(m2, d_m2)
    assert np.array_equal(i2, ids) and np.array_equal(m2, mask)
    # 16-bit entries (the bundled vocabulary's ids fit): half the bytes, same rows back
    total16 = ctx.compact_rows(d_ids, d_nr, n, L, d_comp, bits=16)
    assert total16 == total
    comp16 = np.empty(total, dtype=np.uint16); ctx.d2h(comp16, d_comp)
    assert np.array_equal(comp16.astype(np.int32), comp)
    ctx.expand_rows(d_comp, d_nr, n, L, d_i2, d_m2, bits=16)
    ctx.sync()
    ctx.d2h(i2, d_i2); ctx.d2h(m2, d_m2)
    assert np.array_equal(i2, ids) and np.array_equal(m2, mask)
    for p in (d_ids, d_nr, d_comp, d_i2, d_m2):
        ctx.free(p)


def test_compact16_refuses_wide_ids(sampler, tmp_path):
    """A vocabulary with ids above 65535 cannot use the 16-bit exchange form."""
    from genz_tokenize import Tokenize, _native
    v, b = corpus.custom_tables()                      # 100 k entries
    (tmp_path / "v").write_bytes(v); (tmp_path / "b").write_bytes(b)
    t = Tokenize.fromFile(str(tmp_path / "v"), str(tmp_path / "b"))
    t._sync_tables()
    ctx = t._ctx
    d = ctx.alloc(1024)
    with pytest.raises(_native.GzError) as e:
        ctx.compact_rows(d, d, 1, 4, d, bits=16)
    assert e.value.code == _native.GZ_E_LIMIT
    ctx.free(d)


def test_g6_decode_batch(tok, tmp_path):
    """decode_batch (GPU) against the reference's decode outputs: '@@ ' joining, unknown ids, custom unk, tables whose
    words contain spaces / '@@ ' / nothing at all, id collisions."""
    import base64
    from genz_tokenize import Tokenize
    for row in read_jsonl("g6_decode.jsonl"):
        if "vocab_b64" in row:
            (tmp_path / "v").write_bytes(base64.b64decode(row["vocab_b64"]))
            (tmp_path / "b").write_bytes(base64.b64decode(row["bpe_b64"]))
            t = Tokenize.fromFile(str(tmp_path / "v"), str(tmp_path / "b"))
        elif row["unk_token"] != "<unk>":
            t = Tokenize(unk_token=row["unk_token"])
        else:
            t = tok
        assert t.decode_batch(row["ids"]) == row["result"], row["name"]
        for ids, want in list(zip(row["ids"], row["result"]))[:40]:
            assert t.decode(ids) == want                      # the single-call form (host dict, like the reference)
            assert t.decode_batch([ids]) == [want]
    assert tok.decode_batch([]) == []
    assert tok.decode_batch([[], []]) == ["", ""]


def test_encode_decode_round_trip(tok, oracle_tables, sampler):
    """20 k documents: encode on the GPU, decode the dense [N, L] id matrix on the GPU, compare every row with the
    oracle's decode of the same ids; for documents without unknown pieces the text comes back word for word."""
    text, offs, L = corpus.config_corpus(3, n_docs=20000, seed=41, sampler=sampler)
    out = tok.encode_packed(text, offs, max_len=L)
    ids = out["input_ids"]
    got = tok.decode_batch(ids)
    raw = text.tobytes()
    exact = 0
    for i in range(0, len(got), 7):
        assert got[i] == O.decode(ids[i].tolist(), oracle_tables), i
    unk = oracle_tables.unk_id
    for i in range(len(got)):
        n = int(out["n_real"][i])
        if n < L and unk not in ids[i, :n]:
            doc = raw[offs[i]:offs[i + 1]].decode("utf-8")
            body = got[i].split(" </s>")[0][len("<s> "):] if n > 2 else ""
            assert body.split() == doc.split(), i
            exact += 1
    assert exact > 10000


def test_g7_preprocess(oracle_tables):
    """The text pre-pass (remove_html / convert_unicode / remove_punctuations / remove_emoji / remove_URL) on the GPU
    against the reference's outputs, one filter at a time and chained; batched by filter chain."""
    from genz_tokenize import preprocess as P
    rows = read_jsonl("g7_preprocess.jsonl.gz")
    groups = {}
    for r in rows:
        groups.setdefault(tuple(r["ops"]), []).append(r)
    assert len(groups) >= 13
    for ops, rs in groups.items():
        got = P.preprocess_batch([r["text"] for r in rs], list(ops))
        for r, g in zip(rs, got):
            assert g == r["result"], (ops, r["text"], g)
    # the single-call drop-in names
    for r in rows[:300]:
        fn = {"html": P.remove_html, "unicode": P.convert_unicode, "punct": P.remove_punctuations, "emoji": P.remove_emoji,
              "url": P.remove_URL}[r["ops"][0]]
        assert fn(r["text"]) == r["result"]
    assert P.preprocess_batch([], ["html"]) == []
    with pytest.raises(TypeError):
        P.remove_html(None)


def test_preprocess_long_documents_vs_oracle():
    """Documents far longer than a tile: tags, URLs and whitespace runs that straddle tile boundaries, open tags that
    never close, and every filter's carried state."""
    import random
    from genz_tokenize import preprocess as P
    r = random.Random(77)
    parts = ["<b>", "</b>", "<a href='http://x.vn/", "'>", "<", ">", "http://vnexpress.net/" + "a" * 90, "https", "http", " ", "  ", "\n",
             "\u3000", "\u00a0", "\u2003", "\U0001F600", "\u2764\ufe0f", "\u4e2d\u6587", "a\u0300", "\u00e2\u0301", "\u01b0\u0323", "Y\u0309",
             "x\u0301", "\u0300", "vi\u1ec7t", "nam", "!?.,", "@#", "\u1ee9ng_d\u1ee5ng", "e" * 70, "<" + "q" * 130 + ">", "\t"]
    docs = []
    for n in [1, 5, 63, 64, 65, 127, 128, 129, 500, 2000, 9000]:
        for _ in range(6):
            docs.append("".join(r.choice(parts) for _ in range(n)))
    docs += ["<" + "a" * 300, "a" * 64 + ">" + "<" * 64, "http" + "x" * 200 + " y", " " * 200 + "a" + " " * 200, "", "\U0001F600" * 100,
             ("a\u0300" * 40), "x" * 63 + "a\u0300", "x" * 62 + "\u00e2\u0301", "x" * 63 + "\u00e2\u0301", "x" * 61 + "http://a b"]
    for ops in (["html"], ["unicode"], ["punct"], ["emoji"], ["url"], ["html", "url", "unicode", "emoji", "punct"], ["url", "html"]):
        got = P.preprocess_batch(docs, ops)
        for d, g in zip(docs, got):
            assert g == O.preprocess(d, ops), (ops, d[:80])


def test_preprocess_character_classes_code_point_by_code_point():
    """The punctuation + emoji pass takes its character classes (str.isspace, string.punctuation, the emoji ranges of
    preprocess.py:51-70) from bit-mask lookups and decodes without a branch: EVERY code point up to U+3100, the emoji planes'
    edges and the ends of the UTF-8 length classes, each between two letters (kept / deleted / turned into one space), against the
    oracle -- alone, merged with remove_punctuations, and through the other filters."""
    from genz_tokenize import preprocess as P
    cps = [c for c in range(1, 0x3100) if not 0xD800 <= c <= 0xDFFF]
    cps += [0xFFFF, 0x10000, 0x1F5FF, 0x1F600, 0x1F64F, 0x1F650, 0x1FAFF, 0x2FFFF, 0xE0001, 0x10FFFF, 0xFE0F, 0xFFFD, 0xD7FF, 0xE000]
    docs = ["a" + chr(c) + "b" for c in cps]
    docs += ["".join(chr(c) for c in cps[k:k + 50]) for k in range(0, len(cps), 50)]                # runs: carried states across classes
    docs += ["x" * k + chr(c) + "y" for k in (61, 62, 63) for c in (0x20, 0x85, 0xA0, 0x1680, 0x2003, 0x3000, 0x2764, 0x1F600, 0x21)]   # across a 64-byte step
    for ops in (["emoji"], ["punct", "emoji"], ["punct"], ["html", "unicode", "punct", "emoji", "url"]):
        got = P.preprocess_batch(docs, ops)
        for d, g in zip(docs, got):
            assert g == O.preprocess(d, ops), (ops, [hex(ord(ch)) for ch in d[:6]], g)


def test_preprocess_fused_and_filter_by_filter_paths():
    """Documents of at most 4 096 bytes run their whole filter chain in one kernel, on chip (gz_pp_fused_kernel, with filters
    skipped when a look at the bytes shows they cannot apply); longer ones go filter by filter through HBM.  Both against the
    oracle around the size limit -- 4 094 ... 4 098 bytes with tags, URLs, combining marks, emoji and whitespace runs at the
    very end -- mixed in one batch; and the reference-generated G7 rows + the long-document comparison once more with
    pp_fused = 0 (everything filter by filter) in a child process."""
    import random
    import subprocess
    import sys
    from genz_tokenize import preprocess as P
    r = random.Random(5)
    tails = ["<b>x</b>", "http://a.vn/b c", "a\u0300", "\u00e2\u0301", "\U0001F600 ", "  \u3000 ", "<unclosed", "x!?", "https", "\u0300"]
    fill = ["vi\u1ec7t ", "nam ", "a", " ", "<i>", "</i>", "!", "http://x ", "e\u0301", "\u2764"]
    docs = []
    for size in (2048, 4094, 4095, 4096, 4097, 4098, 9000):
        for t in tails:
            body = ""
            while len((body + t).encode("utf-8")) < size - 8:
                body += r.choice(fill)
            d = body + t
            pad = size - len(d.encode("utf-8"))
            docs.append("x" * max(pad, 0) + d)
            assert abs(len(docs[-1].encode("utf-8")) - size) <= 8
    docs += ["", "a", "<", "http", "a" * 4096, "<" * 4096, " " * 4095 + "a", " " * 4096 + "a"]
    for ops in (["html", "unicode", "punct", "emoji", "url"], ["url", "emoji"], ["emoji"], ["html"], ["unicode", "html"]):
        got = P.preprocess_batch(docs, ops)
        for d, g in zip(docs, got):
            assert g == O.preprocess(d, ops), (ops, len(d.encode("utf-8")), d[-40:])
    env = dict(os.environ, GZ_TEST_SWITCHES="pp_fused=0")
    _run_selection("preprocess_filter_by_filter", env, "g7_preprocess or preprocess_long_documents")


def test_preprocess_device_size_query_and_small_capacity(tok, sampler):
    """The pre-pass makes ONE synchronisation: the packed text is written before the host knows its size.  A capacity that is
    too small must still be honoured on the device (nothing written behind it, GZ_E_CAPACITY afterwards), a call without an
    output buffer returns the size and the offsets, and an empty batch returns [0]."""
    from genz_tokenize import _native
    ctx = tok._ctx
    text, offs, L = corpus.config_corpus(2, n_docs=3000, seed=7, sampler=sampler)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    n, nbytes = len(offs) - 1, int(offs[-1])
    d_text = ctx.alloc(len(text) + 64); ctx.h2d(d_text, text)
    d_off = ctx.alloc(8 * (n + 1)); ctx.h2d(d_off, offs)
    d_oo = ctx.alloc(8 * (n + 1))
    ops = [3, 4]                                                       # punct, emoji: every document shrinks or stays
    total = ctx.preprocess_device(ops, d_text, d_off, n, nbytes, 0, 0, d_oo)           # size query
    oo = np.empty(n + 1, np.int64); ctx.d2h(oo, d_oo)
    assert 0 < total < nbytes and oo[0] == 0 and oo[-1] == total and np.all(np.diff(oo) >= 0)
    d_out = ctx.alloc(nbytes + 64)
    assert ctx.preprocess_device(ops, d_text, d_off, n, nbytes, d_out, total, d_oo) == total   # exactly enough
    full = np.empty(total, np.uint8); ctx.d2h(full, d_out)
    cap = int(oo[n // 2]) + 5                                          # ends inside a document
    guard = np.full(nbytes + 64, 0xA5, np.uint8); ctx.h2d(d_out, guard)
    with pytest.raises(_native.GzError) as e:
        ctx.preprocess_device(ops, d_text, d_off, n, nbytes, d_out, cap, d_oo)
    assert e.value.code == _native.GZ_E_CAPACITY
    got = np.empty(nbytes + 64, np.uint8); ctx.d2h(got, d_out)
    assert np.all(got[cap:] == 0xA5)                                   # nothing behind the capacity
    k = int(np.searchsorted(oo, cap, side="right")) - 1                # documents that end within it were written
    assert np.array_equal(got[:oo[k]], full[:oo[k]])
    assert ctx.preprocess_device(ops, d_text, d_off, 0, 0, d_out, 16, d_oo) == 0
    z = np.full(1, -1, np.int64); ctx.d2h(z, d_oo)
    assert z[0] == 0
    for b in (d_text, d_off, d_oo, d_out):
        ctx.free(b)


@pytest.mark.parametrize("n", [16383, 16384, 16385, 18432, 18433, 40000])
def test_row_offsets_of_large_ragged_batches(tok, n):
    """Ragged layouts, batch decode and the pre-pass's general tail get their offsets from a 64-bit scan that runs on many
    workgroups from 16 384 rows on (gz_scan64_*: block totals travel in the next block's first slot): sizes around the switch
    and around its 2 048-row blocks, rows of seven different lengths (empty documents among them), offsets and contents against
    the oracle; the same rows back through batch decode."""
    kinds = ["", "a", "vi\u1ec7t nam", "xin ch\u00e0o c\u00e1c b\u1ea1n", "h\u1ecdc sinh gi\u1ecfi", "zzqqx", "c\u00f4ng_ngh\u1ec7 th\u00f4ng_tin 2024"]
    docs = [kinds[(i * 7 + i // 3) % len(kinds)] for i in range(n)]
    raw = [d.encode("utf-8") for d in docs]
    offs = np.zeros(n + 1, np.int64); np.cumsum([len(b) for b in raw], out=offs[1:])
    text = np.frombuffer(b"".join(raw), dtype=np.uint8)
    got = tok.encode_packed(text, offs, max_len=None)
    ro = np.asarray(got["row_off"], dtype=np.int64)
    ids = np.asarray(got["input_ids"]).reshape(-1)
    single = {k: tok(k)["input_ids"] for k in kinds}                  # (the single-call path is checked against the reference elsewhere)
    lens = np.array([len(single[d]) for d in docs], dtype=np.int64)
    assert ro[0] == 0 and np.array_equal(np.diff(ro), lens)
    assert ro[-1] == len(ids)
    for i in list(range(0, n, 997)) + [2047, 2048, 2049, 16383, n - 1]:
        if i < n:
            assert ids[ro[i]:ro[i + 1]].tolist() == single[docs[i]], i
    back = tok.decode_batch([ids[ro[i]:ro[i + 1]].tolist() for i in range(n)])
    ref = {k: tok.decode(single[k]) for k in kinds}
    assert back == [ref[d] for d in docs]


def test_device_handoff_dlpack():
    """encode_to_device keeps the [N, L] outputs in HBM; torch.from_dlpack reads them zero-copy.  Runs in a child
    process because torch must initialise its GPU context BEFORE this library is loaded (both resolve the HIP runtime
    by soname; the first one loaded serves both) -- see tests/handoff_child.py."""
    import subprocess, sys
    pytest.importorskip("torch")
    r = _run_child([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "handoff_child.py")], "handoff", timeout=300)
    assert r.returncode == 0 and "HANDOFF OK" in r.stdout, r.report


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_random_tables_fuzz(seed, tmp_path):
    """Random merge tables over a tiny alphabet (deep merge chains, overlapping pairs like 'a a', the same string
    reachable through different merges, literal '</w>' / '@@' pieces, missing vocabulary entries) and random texts with
    words of 1..300 symbols, against the plain-C restatement -- every merge kernel width, both word-table modes."""
    import random
    import gz_oracle_c as OC
    from genz_tokenize import Tokenize
    r = random.Random(1000 + seed)
    alphabet = list("ab") + r.sample(list("cdeghk_"), 3) + r.sample(["\u00e2", "\u1ec7", "\u0111", "\U0001F600", "\u4e2d"], 2)
    if seed % 2 == 0:
        alphabet += ["<", "/", "w", ">", "@"]
    syms = list(alphabet)
    finals = [a + "</w>" for a in alphabet]
    merges = []
    for _ in range(r.choice([40, 150, 600, 1500])):
        left = r.choice(syms)
        if r.random() < 0.35:
            right = r.choice(finals); new = left + right; finals.append(new)
        else:
            right = r.choice(syms); new = left + right; syms.append(new)
        if len(new) > 40:
            continue
        merges.append(left + " " + right)
    if r.random() < 0.5:
        merges.insert(0, "#version: 0.2")
    if seed == 3:
        merges += merges[5:25]                                   # duplicate lines: the later rank wins
    bpe = ("\n".join(merges) + "\n").encode("utf-8")
    words = set()
    for x in syms:
        if r.random() < 0.8: words.add(x + "@@")
    for x in finals:
        if r.random() < 0.8: words.add(x[:-4])
    words = sorted(words); r.shuffle(words)
    vocab = "".join("%s %d\n" % (w, r.randint(1, 99)) for w in words).encode("utf-8")
    (tmp_path / "v").write_bytes(vocab); (tmp_path / "b").write_bytes(bpe)
    tok = Tokenize.fromFile(str(tmp_path / "v"), str(tmp_path / "b"))
    co = OC.COracle(vocab, bpe)

    def word():
        n = r.choice([1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 17, 20, 31, 32, 33, 50, 64, 65, 100, 300])
        return "".join(r.choice(alphabet) for _ in range(n))
    docs = []
    for _ in range(1500):
        k = r.choice([0, 1, 2, 5, 10, 30])
        docs.append("".join(word() + r.choice([" ", " ", "  ", "\n", "\n ", "\t", "\u3000"]) for _ in range(k)))
    text, toff = OC._pack(docs)
    text = np.ascontiguousarray(text)                        # (one spare byte at the end; offsets do not cover it)
    for ml, pad, tr in ((None, True, True), (24, True, True)):
        ids, mask, _, _, row, _, _ = co.call_packed(text, toff, max_len=ml, padding=pad, truncation=tr)
        lens = np.diff(row)
        for wt in (True, False):
            out = tok.encode_packed(text, toff, max_len=ml, padding=pad, truncation=tr, word_table=wt)
            ro = np.asarray(out["row_off"], np.int64)
            assert np.array_equal(np.diff(ro), lens), (seed, ml, wt)
            got = _gather(out["input_ids"].reshape(-1), ro[:-1], lens)
            if not np.array_equal(got, ids[:row[-1]]):
                bad = int(np.nonzero(got != ids[:row[-1]])[0][0])
                d = int(np.searchsorted(row, bad, side="right") - 1)
                raise AssertionError((seed, ml, wt, "doc", d, docs[d][:200], got[row[d]:row[d + 1]][:40], ids[row[d]:row[d + 1]][:40]))
            assert np.array_equal(_gather(out["attention_mask"].reshape(-1), ro[:-1], lens), mask[:row[-1]])


def test_extreme_batch_shapes(tok, sampler):
    """Batches far from the benchmark's shape, against the plain-C restatement: a few multi-megabyte documents next to
    tiny ones (one wave assembles a 600 k-token row), tens of thousands of empty documents, and a batch that is
    nothing but empty strings."""
    import gz_oracle_c as OC
    from corpus import VOCAB_PATH, BPE_PATH
    co = OC.COracle(open(VOCAB_PATH, "rb").read(), open(BPE_PATH, "rb").read())
    text, offs, _ = corpus.config_corpus(3, n_docs=30000, seed=51, sampler=sampler)
    raw = text.tobytes()
    big1, big2 = raw[:offs[12000]].decode("utf-8"), raw[offs[12000]:offs[21000]].decode("utf-8")
    docs = ["xin chào", big1, "", "a", big2, "\n\n", "việt nam " * 3] + [""] * 40000 + ["hết"]
    t, to = OC._pack(docs)
    t = np.ascontiguousarray(t)
    for ml, pad, tr in ((None, True, True), (300, True, True)):
        ids, mask, _, _, row, _, _ = co.call_packed(t, to, max_len=ml, padding=pad, truncation=tr)
        out = tok.encode_packed(t, to, max_len=ml, padding=pad, truncation=tr)
        ro = np.asarray(out["row_off"], np.int64)
        lens = np.diff(row)
        assert np.array_equal(np.diff(ro), lens)
        assert np.array_equal(_gather(out["input_ids"].reshape(-1), ro[:-1], lens), ids[:row[-1]])
        assert np.array_equal(_gather(out["attention_mask"].reshape(-1), ro[:-1], lens), mask[:row[-1]])
    empties = [""] * 5000
    out = tok.encode_batch(empties, max_len=8)
    sp = tok._special_ids()
    assert np.array_equal(out["input_ids"], np.tile(np.array([sp[1], sp[2]] + [sp[0]] * 6, dtype=np.int32), (5000, 1)))
    out = tok.encode_batch(empties, empties, max_len=None)
    assert np.array_equal(np.diff(out["row_off"]), np.full(5000, 4))


def test_unpadded_rows_are_counted_then_written_once(tok, sampler):
    """Single texts without padding (max_len None, padding False) take gz_rowsr_kernel: a count pass, the scan, then every row
    written once at its place.  Documents around both buffer sizes of its rounds (512 and 1 536 words / tokens), empty ones, ones
    whose words all go through the merge loop (many pieces per word: more tokens than words), and ones far too long for a round
    (the plain walk) -- in a batch of short documents (8 per wave, small buffers) and in a batch of long ones (1 per wave, large
    buffers); against the plain-C restatement, with the whole-word tables on and off; n_real; and a capacity that is too small."""
    import random
    import gz_oracle_c as OC
    from genz_tokenize import _native
    from corpus import VOCAB_PATH, BPE_PATH
    co = OC.COracle(open(VOCAB_PATH, "rb").read(), open(BPE_PATH, "rb").read())
    r = random.Random(77)
    text, offs, _ = corpus.config_corpus(3, n_docs=6000, seed=61, sampler=sampler)
    raw = text.tobytes()
    base = [raw[offs[i]:offs[i + 1]].decode("utf-8") for i in range(6000)]
    words = " ".join(base[:400]).split()
    def doc(nw, odd=0.0):
        ws = [(w if r.random() >= odd else "".join(r.choice("qxzwkjfđâệ") for _ in range(r.choice([3, 7, 12, 18])))) for w in (r.choice(words) for _ in range(nw))]
        return " ".join(ws)
    special = [doc(n, o) for n in (0, 1, 62, 63, 64, 65, 300, 480, 505, 510, 511, 512, 513, 600, 1000, 1500, 1530, 1534, 1535, 1536, 1537, 1600, 4000, 9000)
               for o in (0.0, 0.5, 1.0)]
    for kind in ("short", "long"):
        docs = list(base[:3000]) if kind == "short" else [doc(r.choice([200, 400, 700, 900]), r.choice([0.0, 0.1])) for _ in range(600)]
        for k, d in enumerate(special):
            docs.insert((k * 37) % len(docs), d)
        docs += ["", "", "a", ""]
        t, to = OC._pack(docs)
        t = np.ascontiguousarray(t); to = np.ascontiguousarray(to, dtype=np.int64)
        assert (kind == "long") == (int(to[-1]) // len(docs) > 1024)
        for ml, pad, tr in ((None, True, True), (50, False, True)):
            ids, mask, _, _, row, _, _ = co.call_packed(t, to, max_len=ml, padding=pad, truncation=tr)
            lens = np.diff(row)
            for wt in (True, False):
                out = tok.encode_packed(t, to, max_len=ml, padding=pad, truncation=tr, word_table=wt)
                ro = np.asarray(out["row_off"], np.int64)
                assert np.array_equal(np.diff(ro), lens)
                assert np.array_equal(out["input_ids"].reshape(-1)[:row[-1]], ids[:row[-1]])
                assert np.array_equal(out["attention_mask"].reshape(-1)[:row[-1]], mask[:row[-1]])
                assert np.array_equal(np.asarray(out["n_real"]), lens)
        # a capacity one entry short: GZ_E_CAPACITY, nothing written
        ctx = tok._ctx
        n = len(docs); total = int(row[-1])
        d_t = ctx.alloc(len(t) + 64); ctx.h2d(d_t, t)
        d_o = ctx.alloc(8 * (n + 1)); ctx.h2d(d_o, to)
        d_i, d_m, d_r, d_ro = ctx.alloc(4 * total), ctx.alloc(4 * total), ctx.alloc(4 * n), ctx.alloc(8 * (n + 1))
        fill = np.full(total, 0x5A5A5A5A, dtype=np.int32)
        ctx.h2d(d_i, fill); ctx.h2d(d_m, fill)
        with pytest.raises(_native.GzError) as e:
            ctx.encode_device(d_t, d_o, 0, 0, n, 0, _native.GZ_MAX_LEN_NONE, total - 1, d_i, d_m, d_row_off=d_ro, d_n_real=d_r, h_text_off=to)
            ctx.sync()
        assert e.value.code == _native.GZ_E_CAPACITY
        got = np.empty(total, np.int32); ctx.d2h(got, d_i)
        assert np.all(got == 0x5A5A5A5A)
        ctx.encode_device(d_t, d_o, 0, 0, n, 0, _native.GZ_MAX_LEN_NONE, total, d_i, d_m, d_row_off=d_ro, d_n_real=d_r, h_text_off=to)
        ctx.sync()
        ctx.d2h(got, d_i)
        assert np.array_equal(got, ids[:total])
        for q in (d_t, d_o, d_i, d_m, d_r, d_ro):
            ctx.free(q)
    # the one-launch kernel places the rows of a call that ONE workgroup holds (layout 2): the same capacity rule there, single and paired
    docs = ["xin chào việt nam", "", "sinh_viên công_nghệ qxzwk"]
    t, to = OC._pack(docs)
    t = np.ascontiguousarray(t); to = np.ascontiguousarray(to, dtype=np.int64)
    for pair in (False, True):
        ids, mask, _, _, row, _, _ = co.call_packed(t, to, t if pair else None, to if pair else None, None, True, True)
        total = int(row[-1])
        out = tok.encode_packed(t, to, t, to, max_len=None) if pair else tok.encode_packed(t, to, max_len=None)
        assert np.array_equal(np.asarray(out["row_off"], np.int64), row) and np.array_equal(out["input_ids"].reshape(-1)[:total], ids[:total])
        assert np.array_equal(out["attention_mask"].reshape(-1)[:total], mask[:total])
        n = len(docs)
        d_t = ctx.alloc(len(t) + 64); ctx.h2d(d_t, t)
        d_o = ctx.alloc(8 * (n + 1)); ctx.h2d(d_o, to)
        bufs = [ctx.alloc(4 * total + 64) for _ in range(4)] + [ctx.alloc(8 * (n + 1)), ctx.alloc(8 * n + 8), ctx.alloc(4 * n + 4), ctx.alloc(4 * n + 4)]
        d_i, d_m, d_tt, d_sq, d_ro, d_pl, d_r, d_st = bufs
        fill = np.full(total, 0x5A5A5A5A, dtype=np.int32)
        ctx.h2d(d_i, fill)
        kw = dict(d_tt=d_tt, d_seq=d_sq, d_pair_len=d_pl, d_status=d_st, h_pair_off=to) if pair else {}
        with pytest.raises(_native.GzError) as e:
            ctx.encode_device(d_t, d_o, d_t if pair else 0, d_o if pair else 0, n, 0, _native.GZ_MAX_LEN_NONE, total - 1, d_i, d_m,
                              d_row_off=d_ro, d_n_real=d_r, h_text_off=to, **kw)
            ctx.sync()
        assert e.value.code == _native.GZ_E_CAPACITY
        got = np.empty(total, np.int32); ctx.d2h(got, d_i)
        assert np.all(got == 0x5A5A5A5A)
        for q in [d_t, d_o] + bufs:
            ctx.free(q)


def test_chained_device_calls(tok, sampler):
    """Dense calls with host offsets are enqueued behind one another without a host sync (same workspace, stream order):
    five different batches of growing and shrinking size, chained, must each equal their stand-alone result."""
    from genz_tokenize import _native
    ctx = tok._ctx
    flags = _native.GZ_PADDING | _native.GZ_TRUNCATION | _native.GZ_TIMING
    jobs = []
    for seed, n, L in ((1, 3000, 64), (2, 20000, 48), (3, 50, 256), (4, 12000, 128), (5, 1, 16)):
        text, offs, _ = corpus.config_corpus(3, n_docs=n, seed=200 + seed, sampler=sampler)
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        want = tok.encode_packed(text, offs, max_len=L)
        d_t = ctx.alloc(len(text) + 64); ctx.h2d(d_t, text)
        d_o = ctx.alloc(8 * (n + 1)); ctx.h2d(d_o, offs)
        d_i, d_m, d_r = ctx.alloc(4 * n * L), ctx.alloc(4 * n * L), ctx.alloc(4 * n)
        jobs.append((text, offs, n, L, want, d_t, d_o, d_i, d_m, d_r))
    ctx.sync()
    for _, offs, n, L, _, d_t, d_o, d_i, d_m, d_r in jobs:
        ctx.encode_device(d_t, d_o, 0, 0, n, L, flags, n * L, d_i, d_m, d_n_real=d_r, h_text_off=offs)      # no sync in between
    hist = ctx.timing_history(64)
    assert len(hist) == len(jobs) and all(t > 0 for t in hist)
    for _, offs, n, L, want, d_t, d_o, d_i, d_m, d_r in jobs:
        ids = np.empty((n, L), np.int32); mask = np.empty((n, L), np.int32); nr = np.empty(n, np.int32)
        ctx.d2h(ids, d_i); ctx.d2h(mask, d_m); ctx.d2h(nr, d_r)
        assert np.array_equal(ids, want["input_ids"]) and np.array_equal(mask, want["attention_mask"])
        assert np.array_equal(nr, want["n_real"])
        for p in (d_t, d_o, d_i, d_m, d_r):
            ctx.free(p)
