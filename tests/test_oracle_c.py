"""The plain-C restatement (oracle/gz_oracle.c) against the reference's golden vectors and against the Python
restatement.  CPU only.  Both oracles are test infrastructure; the product never loads either."""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

import gz_oracle as O
import gz_oracle_c as OC
from conftest import GOLDEN, ROOT, read_jsonl

DATA = os.path.join(ROOT, "genz-tokenize_amd", "genz_tokenize", "data")


@pytest.fixture(scope="module")
def c_oracle():
    return OC.COracle(open(os.path.join(DATA, "vocab.txt"), "rb").read(), open(os.path.join(DATA, "bpe.codes"), "rb").read())


def _call_rows(rows):
    """Golden `call` rows the C face can express: str arguments (TypeError rows are host-side checks)."""
    out = []
    for r in rows:
        if r.get("kind", "call") != "call" or any(r.get("bytes_args", [])) or r.get("raises") not in (None, "ValueError"):
            continue
        if not all(isinstance(a, str) for a in r["args"]):
            continue
        out.append(r)
    return out


def _check_rows(co, rows):
    groups = {}
    for r in rows:
        kw = r["kwargs"]
        key = (len(r["args"]), kw.get("max_len"), kw.get("padding", True), kw.get("truncation", True))
        groups.setdefault(key, []).append(r)
    n = 0
    for (na, ml, pad, tr), rs in groups.items():
        texts = [r["args"][0] for r in rs]
        pairs = [r["args"][1] for r in rs] if na == 2 else None
        I, M, T, S, st = co.call_batch(texts, pairs, ml, pad, tr)
        for k, r in enumerate(rs):
            if "raises" in r:
                assert st[k] == 1, r
                continue
            assert st[k] == 0, r
            res = r["result"]
            assert I[k] == res["input_ids"] and M[k] == res["attention_mask"], r
            if na == 2:
                none = lambda xs: [OC_NONE if v is None else v for v in xs]
                assert T[k] == none(res["token_type_ids"]) and S[k] == none(res["sequence_id"]), r
            n += 1
    return n


OC_NONE = O.NONE


def test_c_g1(c_oracle):
    rows = _call_rows(read_jsonl("g1_cases.jsonl"))
    assert len(rows) >= 40
    assert _check_rows(c_oracle, rows) >= 30
    for r in read_jsonl("g1_cases.jsonl"):
        if r["kind"] == "vocab_size":
            assert c_oracle.vocab_size == r["result"]


def test_c_g3(c_oracle):
    rows = _call_rows(read_jsonl("g3_random.jsonl.gz"))
    assert len(rows) == 2400
    assert _check_rows(c_oracle, rows) >= 2000


def test_c_g4_loader():
    for row in read_jsonl("g4_loader.jsonl"):
        v, b = base64.b64decode(row["vocab_b64"]), base64.b64decode(row["bpe_b64"])
        if "calls" not in row:
            if row["raises"] is None:
                OC.COracle(v, b)
            else:
                with pytest.raises(UnicodeDecodeError):
                    OC.COracle(v, b)
            continue
        co = OC.COracle(v, b)
        assert co.vocab_size == row["vocab_size"], row["name"]
        assert co.n_ranks == len(row["bpe_ranks"]), row["name"]
        for tok, i in row["encoder"]:
            assert co.lookup(tok) == i, (row["name"], tok)
        _check_rows(co, _call_rows(row["calls"]))


def _hash_blocks(co, text, offs, L, block):
    hi, hm, ntok = [], [], 0
    n = len(offs) - 1
    for lo in range(0, n, block):
        o = offs[lo:min(lo + block, n) + 1]
        ids, mask, _, _, row, _, _ = co.call_packed(text, o, max_len=L)
        k = int(row[-1])
        hi.append(hashlib.sha256(ids[:k].astype("<i4").tobytes()).hexdigest())
        hm.append(hashlib.sha256(mask[:k].astype("<i4").tobytes()).hexdigest())
        ntok += int(mask[:k].sum())
    return hi, hm, ntok


@pytest.mark.parametrize("name", ["cfg2_10k", "cfg3_20k", "cfg5_300", "cfg3_1M"])
def test_c_g5_full(c_oracle, name):
    """Every block of the reference-hashed corpora (the Python oracle only affords the first block), up to the
    full 1M-document headline workload (about a minute)."""
    import corpus
    e = json.load(open(os.path.join(GOLDEN, "g5_hashes.json")))[name]
    s = corpus.Sampler()
    text, offs, L = corpus.config_corpus(e["cfg"], n_docs=e["n_docs"], sampler=s)
    co = c_oracle
    if e["custom_tables"]:
        v, b = corpus.custom_tables()
        co = OC.COracle(v, b)
    hi, hm, ntok = _hash_blocks(co, np.ascontiguousarray(text), np.ascontiguousarray(offs, dtype=np.int64), L, e["block"])
    assert hi == e["ids_sha256"] and hm == e["mask_sha256"] and ntok == e["n_tokens"]


def test_c_cfg4_shard_prefix_matches_the_reference(c_oracle):
    """The per-shard digests of BASELINE configs[3] come from the C oracle (whole shard) AND from the reference (first
    20 000 documents of the same shard): here the C oracle is run on that prefix of one shard and must reproduce the
    reference's digests, and its first whole-shard block must reproduce the committed whole-shard digest."""
    import corpus
    e = json.load(open(os.path.join(GOLDEN, "g5_hashes.json")))["cfg4_shard5"]
    text, offs, L = corpus.config_corpus(4, n_docs=e["n_docs"], seed=e["seed"])
    text = np.ascontiguousarray(text); offs = np.ascontiguousarray(offs, dtype=np.int64)
    assert int(offs[-1]) == e["input_bytes"]
    r = e["ref_prefix"]
    hi, hm, ntok = _hash_blocks(c_oracle, text, offs[:r["n_docs"] + 1], L, r["block"])
    assert hi == r["ids_sha256"] and hm == r["mask_sha256"] and ntok == r["n_tokens"]
    hi, hm, _ = _hash_blocks(c_oracle, text, offs[:e["block"] + 1], L, e["block"])
    assert hi[0] == e["ids_sha256"][0] and hm[0] == e["mask_sha256"][0]


def test_c_cfg5_50k_prefix_matches_the_reference():
    """BASELINE configs[4] at full size: the C oracle on the first 300 of the 50 000 documents (custom tables) against
    the reference's digests of the same documents, padded and unpadded."""
    import corpus
    e = json.load(open(os.path.join(GOLDEN, "g5_hashes.json")))["cfg5_50k"]
    text, offs, L = corpus.config_corpus(5, n_docs=e["n_docs"])
    text = np.ascontiguousarray(text); offs = np.ascontiguousarray(offs, dtype=np.int64)
    assert int(offs[-1]) == e["input_bytes"]
    v, b = corpus.custom_tables()
    co = OC.COracle(v, b)
    for name in ("padded", "unpadded"):
        r = e[name]["ref_prefix"]
        hi, hm, ntok = _hash_blocks(co, text, offs[:r["n_docs"] + 1], e[name]["max_len"], r["block"])
        assert hi == r["ids_sha256"] and hm == r["mask_sha256"] and ntok == r["n_tokens"], name


@pytest.mark.parametrize("pair", [False, True])
def test_c_vs_python_noisy(c_oracle, oracle_tables, pair):
    import corpus
    s = corpus.Sampler()
    text, offs, _ = corpus.config_corpus(2, n_docs=600, sampler=s, seed=77)
    text, offs = corpus.add_noise(text, offs, seed=5)
    raw = text.tobytes()
    docs = [raw[offs[i]:offs[i + 1]].decode("utf-8", "surrogatepass") for i in range(len(offs) - 1)]
    pairs = docs[::-1] if pair else None
    for ml, pad, tr in [(None, True, True), (48, True, True), (48, True, False), (0, True, True), (-3, True, True), (64, False, True)]:
        assert c_oracle.call_batch(docs, pairs, ml, pad, tr) == tuple(O.call_batch(oracle_tables, docs, pairs, ml, pad, tr))
