#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE.

Run only in the build container (needs /root/reference, read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference package is imported from /root/reference under its own name; the
fixtures written here are data only (inputs + the reference's outputs):

  g1_cases.jsonl      hand-picked calls (README.md:11-15 vector, SURVEY.md §8(a) table,
                      encode / decode / bpe / offsets / TypeError cases), bundled tables
  g3_random.jsonl.gz  2 400 seeded random calls over a hostile alphabet, single + pair,
                      every max_len / padding / truncation combination, exceptions recorded
  g4_loader.jsonl     Tokenize.fromFile on tiny synthetic tables exercising loader rules
                      L1-L7 (file bytes are stored in the fixture, base64)
  g6_decode.jsonl     Tokenize.decode on seeded random id lists (bundled tables, ids of '@@' pieces at the end of a
                      row, out-of-range / negative ids, a non-default unk_token) and on tiny tables whose words
                      contain spaces, '@@ ' and empty strings (file bytes stored base64)
  g7_preprocess.jsonl.gz  the five text filters of genz_tokenize/preprocess.py (remove_html, convert_unicode,
                      remove_punctuations, remove_emoji, remove_URL) on hand-picked and seeded random strings, one
                      filter at a time and chained
  g5_hashes.json      SHA-256 of the reference's input_ids/attention_mask over the seeded
                      synthetic corpora of corpus.py (corpora are re-generated from seed)
"""
import base64
import hashlib
import json
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from genz_tokenize import Tokenize  # noqa: E402  (the reference)
from genz_tokenize.tokenize import get_pairs  # noqa: E402

sys.path.insert(0, ROOT)
import corpus  # noqa: E402
import numpy as np  # noqa: E402


def run(tok, args, kwargs):
    try:
        return {"result": tok(*args, **kwargs)}
    except Exception as e:  # noqa: BLE001
        return {"raises": type(e).__name__, "msg": str(e)}


def jl(path, rows):
    import gzip
    opener = (lambda q: gzip.open(q, "wt", encoding="ascii", compresslevel=9)) if path.endswith(".gz") \
        else (lambda q: open(q, "w", encoding="ascii"))
    with opener(os.path.join(HERE, path)) as f:
        for r in rows:
            f.write(json.dumps(r, ensure_ascii=True) + "\n")
    print(path, len(rows))


# ------------------------------------------------------------------ G1
def g1(tok):
    rows = []
    calls = [
        (("sinh_viên công_nghệ", "hello"), dict(max_len=10, padding=True, truncation=True)),
        (("hello\nworld",), {}), (("",), {}), (("   ",), {}), (("", ""), {}),
        (("a", ""), dict(max_len=8)), (("a b c d e f g h", "x"), dict(max_len=5)),
        (("a b", "x y z w"), dict(max_len=6)), (("a b", "c"), {}), (("a b", "c"), dict(max_len=6)),
        (("a b c d",), dict(max_len=3)), (("a b c d",), dict(max_len=3, truncation=False)),
        (("a b c d",), dict(max_len=3, padding=False)),
        (("<s> </s> <pad> <unk> <mask>",), {}), (("foo@@ bar</w>",), {}),
        (("Trường đại_học Công_nghiệp zzzqqqxx",), {}),
        (("sinh_viên công_nghệ", "hello"), dict(max_len=10, return_offset=True)),
        (("sinh_viên công_nghệ",), dict(return_offset=True)),
        (("một hai\nba  bốn\n\nnăm", "sáu bảy"), dict(return_offset=True)),
        (("x😀y",), {}), (("gh\r\n ef\n\n",), {}), (("\nabc \n def\n",), {}),
        (("a b c\u3000d\x1ce\x85f\u2003g\u200bh",), {}),
        (("a b c",), dict(max_len=0)), (("a b c",), dict(max_len=-1)), (("a b c",), dict(max_len=1)),
        (("a b c",), dict(max_len=2)), (("a b c", "d"), dict(max_len=1)), (("a b c", "d"), dict(max_len=0)),
        (("a", "b"), dict(max_len=4)), (("a", "b"), dict(max_len=5)), (("a", "b"), dict(max_len=6)),
        (("a", "b"), dict(max_len=7)), (("a", "b"), dict(max_len=3)), (("a", "b"), dict(max_len=2)),
        (("", "b"), dict(max_len=8)), (("", ""), dict(max_len=8)), (("", ""), dict(max_len=4)),
        (("", ""), dict(max_len=3)), (("a b", "c"), dict(max_len=6, truncation=False)),
        (("a b c d e", "c"), dict(max_len=6, truncation=False)),
        (("a b", "c"), dict(max_len=16, padding=False)),
        (("aaaa aaaaa aaaaaa",), {}), (("nnnn hhhh nghnghng",), {}),
        (("a" * 70 + " " + "không" * 30,), {}),
        ((None,), {}), ((5,), {}), ((b"abc",), {}), ((["a"],), {}), (("a", 5), {}),
    ]
    for args, kw in calls:
        rows.append({"kind": "call", "args": list(a.decode() if isinstance(a, bytes) else a for a in args),
                     "bytes_args": [isinstance(a, bytes) for a in args], "kwargs": kw, **run(tok, args, kw)})
    for s in ["sinh_viên công_nghệ", "", "hello\nworld", "x😀y"]:
        for ro in (False, True):
            r = tok.encode(s, ro)
            rows.append({"kind": "encode", "text": s, "return_offset": ro,
                         "result": r if not ro else [r[0], r[1]]})
    for ids in ([1, 770, 1444, 2, 0, 0], [99999999], [], [1, 770, 2], [15117, 3019, 4, 13676]):
        rows.append({"kind": "decode", "ids": ids, "result": tok.decode(ids)})
    for w in ["hello", "công_nghệ", "a", "zzzqqqxx", "hello\n", "x😀y", "</w>", "a</w>", "@@", "không"]:
        rows.append({"kind": "bpe", "word": w, "result": tok.bpe(w)})
    try:
        tok.bpe("")
    except Exception as e:  # noqa: BLE001
        rows.append({"kind": "bpe", "word": "", "raises": type(e).__name__})
    for w in [("a", "b", "c"), ("a",), ("a", "a", "a")]:
        rows.append({"kind": "get_pairs", "word": list(w), "result": sorted(list(p) for p in get_pairs(w))})
    # edge rows of the public helpers (tokenize.py:154-182, :270-278): empty / one element / no eos, raises recorded
    def rec(fn, *a):
        try:
            return {"result": fn(*a)}
        except Exception as e:  # noqa: BLE001
            return {"raises": type(e).__name__, "msg": str(e)}
    for w in [(), ("a",), ("a", "b"), ("ab", "ab", "ab", "c"), "abc", ""]:
        r = rec(get_pairs, w)
        if "result" in r:
            r["result"] = sorted(list(q) for q in r["result"])
        rows.append({"kind": "get_pairs_edge", "word": list(w) if isinstance(w, tuple) else w, "is_str": isinstance(w, str), **r})
    for ids in ([], [1], [2], [1, 2], [5, 6, 7], [1, 5, 6], [1, 5, 2], [1, 2, 2, 2], [1, 5, 2, 2, 6, 7, 2, 0, 0],
                [1, 5, 2, 2, 2, 0, 0], [1, 5, 2, 6, 2, 7, 2], [2, 2, 2, 2], [1, 1, 2, 1, 2], [0, 0, 0]):
        rows.append({"kind": "get_sequence_id", "ids": ids, **rec(tok.get_sequence_id, list(ids))})
    for seq in ([], [None], [None, None], [None, 0, None], [None, 0, None, None, 1, None], [0, 0, 0, 0],
                [None, 0, 0, None, None, 1, 1, None, 1, 1], [None, None, None, None], [5, None, 7, None, 9]):
        arg = list(seq)
        r = rec(tok.get_token_type, arg)
        rows.append({"kind": "get_token_type", "seq": seq, "after": arg, **r})      # `after`: the argument as the call left it (in place)
    for ids in ([], [0], [1, 0, 5, 0], [7, 7]):
        rows.append({"kind": "get_atttention_mask", "ids": ids, **rec(tok.get_atttention_mask, list(ids))})
    rows.append({"kind": "vocab_size", "result": tok.vocab_size()})
    rows.append({"kind": "helpers", "ids": [1, 5, 0, 2, 0],
                 "attention_mask": tok.get_atttention_mask([1, 5, 0, 2, 0]),
                 "sequence_id": tok.get_sequence_id([1, 5, 2, 2, 6, 2, 0])})
    jl("g1_cases.jsonl", rows)


# ------------------------------------------------------------------ G3
ALPHA_VI = "aăâbcdđeêghiklmnoôơpqrstuưvxyáàảãạắằẳẵặấầẩẫậéèẻẽẹếềểễệíìỉĩịóòỏõọốồổỗộớờởỡợúùủũụứừửữựýỳỷỹỵ"
ALPHA_MISC = "ABCXYZ0123456789_.,!?()<>/@#-'\"w"
WS = [chr(c) for c in (0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x1C, 0x1D, 0x1E, 0x1F, 0x20, 0x85, 0xA0, 0x1680,
                        0x2000, 0x2001, 0x2002, 0x2003, 0x2004, 0x2005, 0x2006, 0x2007, 0x2008, 0x2009,
                        0x200A, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000)]
# non-whitespace look-alikes and encoding edge cases (ZWSP, combining acute, BOM, NUL, DEL, a lone
# surrogate, NEL-adjacent U+0084/U+0086, U+180E, U+200B-U+200D, U+2060, UTF-8 length boundaries)
ODD = ["\U0001F600", "\U0001F680", "\u4e2d", "\u043a", "\u200b", "\u0301", "\ufeff", "\x00", "\x7f", "\ud800",
       "e\u0301", "\u0084", "\u0086", "\u180e", "\u200c", "\u200d", "\u2060", "\u2027", "\u202a", "\u1681",
       "\u167f", "\u2fff", "\u3001", "@@", "</w>", "<s>", "</s>", "<pad>", "\U0010ffff", "\u07ff", "\u0800",
       "\uffff", "\U00010000", "\x80", "\xa1", "\x9f"]
WORDS = ["không", "là", "sinh_viên", "công_nghệ", "hello", "Việt_Nam", "người", "nghiêng", "trường",
         "đại_học", "a", "b", "nghệ_thuật", "1234", "tp.hcm", "http://x.y/z"]


def rand_text(r):
    mode = r.random()
    if mode < 0.05:
        return ""
    parts = []
    for _ in range(r.choice([1, 1, 2, 3, 5, 8, 12, 20, 40])):
        x = r.random()
        if x < 0.45:
            w = r.choice(WORDS)
        elif x < 0.75:
            w = "".join(r.choice(ALPHA_VI) for _ in range(r.choice([1, 2, 3, 4, 5, 6, 8, 11, 17, 30, 70])))
        elif x < 0.9:
            w = "".join(r.choice(ALPHA_VI + ALPHA_MISC) for _ in range(r.randint(1, 9)))
        else:
            w = "".join(r.choice(ODD + list(ALPHA_MISC)) for _ in range(r.randint(1, 5)))
        parts.append(w)
        g = r.random()
        if g < 0.7:
            parts.append(" ")
        elif g < 0.9:
            parts.append("".join(r.choice(WS) for _ in range(r.randint(1, 3))))
        # else: no gap
    if r.random() < 0.3:
        parts.insert(0, r.choice(WS))
    return "".join(parts)


def g3(tok):
    r = random.Random(20240603)
    rows = []
    for i in range(2400):
        a = rand_text(r)
        pair = r.random() < 0.4
        args = (a, rand_text(r)) if pair else (a,)
        kw = {}
        ml = r.choice([None, None, 1, 2, 3, 4, 5, 6, 8, 16, 32, 128, 0, -2])
        if ml is not None:
            kw["max_len"] = ml
        if r.random() < 0.25:
            kw["padding"] = False
        if r.random() < 0.25:
            kw["truncation"] = False
        if r.random() < 0.15:
            kw["return_offset"] = True
        rows.append({"kind": "call", "args": list(args), "kwargs": kw, **run(tok, args, kw)})
    jl("g3_random.jsonl.gz", rows)


# ------------------------------------------------------------------ G4
def g4():
    tables = {
        "plain": ("a 5\nb 4\nab 3\nab@@ 2\nc 1\nabc 9\naa@@ 1\naa 1\naaa 1\n",
                  "#version: 0.2\na b\na b</w>\nab c</w>\na a\na a</w>\naa a</w>\n"),
        "no_trailing_newline_bpe": ("a 1\nb 1\nab 1\nbc 1\nabc 1\n", "a b\nb c</w>\na bc</w>"),
        "no_header": ("a 1\nb 1\nab 1\nab@@ 1\nc 1\nbc 1\n", "a b\nb c</w>\n"),
        "missing_space_vocab": ("abc\nab 7\n\nb\nx@@ 1\nxy 2\n", "#v\na b\nx y</w>\n"),
        "dup_vocab": ("a 1\nb 1\na 2\nc 1\nd 1\nab 1\n", "#v\na b</w>\n"),
        "special_in_vocab": ("<unk> 1\na 1\n</s> 1\nb 1\n<pad> 5\nc 3\n", "#v\n"),
        "three_field_merge": ("a 1\nb 1\nab 1\nc 1\n", "#v\na b c\na b</w>\n\n  \na\n"),
        "dup_merge": ("a 1\nb@@ 1\nab 1\nb 1\nba 1\nbab 1\nba@@ 1\n", "#v\nb a\na b</w>\nb a\nb a</w>\n"),
        "crlf": ("a 1\r\nb 1\r\nab 1\r\n", "#v\r\na b</w>\r\n"),
        "cr_only": ("a 1\rb 1\rab 1\r", "#v\ra b</w>\r"),
        "bom": ("\ufeffa 1\nb 1\nab 1\n", "\ufeff#v\na b</w>\n"),
        "ws_variants": ("  a 1  \n\tb x 1\nab\u3000 1\n\u2003ab 1\n", "#v\n\u3000a \u00a0b</w> \n"),
        "astral": ("😀 1\n😀@@ 1\nx 1\n😀x 1\nx😀 3\n", "#v\n😀 x</w>\nx 😀</w>\n"),
        "same_string_two_ways": ("abc 1\nab@@ 1\nbc 1\na@@ 1\n", "#v\nb c</w>\na b\na bc</w>\nab c</w>\n"),
        "literal_endmark": ("a 1\n</w> 1\na</w>@@ 1\n</w>@@ 1\nb 1\n",
                            "#v\n< /\n</ w\n</w >\na </w>\n</w> b</w>\n"),
        "empty_files": ("", ""),
        "only_newline": ("\n", "\n"),
    }
    texts = ["a b", "ab", "abc", "aaa aaaa aaaaa", "ab abc c", "a", "b a bab ba", "x y xy",
             "a</w>b", "😀x x😀 😀", "<unk> </s> <pad> a b c", "abc ab bc", "", "ab c\nab\n"]
    rows = []
    tmp = tempfile.mkdtemp()
    for name, (v, b) in tables.items():
        vb, bb = v.encode("utf-8"), b.encode("utf-8")
        vp, bp = os.path.join(tmp, name + ".vocab"), os.path.join(tmp, name + ".bpe")
        open(vp, "wb").write(vb); open(bp, "wb").write(bb)
        try:
            tok = Tokenize.fromFile(vp, bp)
        except Exception as e:  # noqa: BLE001
            rows.append({"name": name, "vocab_b64": base64.b64encode(vb).decode(),
                         "bpe_b64": base64.b64encode(bb).decode(), "raises": type(e).__name__})
            continue
        calls = []
        for t in texts:
            for args, kw in (((t,), {}), ((t, "a b"), {"max_len": 12}), ((t,), {"max_len": 4})):
                calls.append({"args": list(args), "kwargs": kw, **run(tok, args, kw)})
        rows.append({"name": name, "vocab_b64": base64.b64encode(vb).decode(),
                     "bpe_b64": base64.b64encode(bb).decode(),
                     "encoder": sorted(tok.encoder.items(), key=lambda kv: (kv[1], kv[0])),
                     "decoder": sorted(tok.decoder.items()),
                     "bpe_ranks": sorted(([list(k), v] for k, v in tok.bpe_ranks.items()), key=lambda kv: kv[1]),
                     "vocab_size": tok.vocab_size(), "calls": calls})
    # invalid UTF-8 in either file
    for name, vb, bb in (("bad_utf8_vocab", b"a 1\n\xff\xfe 1\n", b"#v\n"), ("bad_utf8_bpe", b"a 1\n", b"#v\n\xc3\n")):
        vp, bp = os.path.join(tmp, name + ".vocab"), os.path.join(tmp, name + ".bpe")
        open(vp, "wb").write(vb); open(bp, "wb").write(bb)
        try:
            Tokenize.fromFile(vp, bp)
            rows.append({"name": name, "vocab_b64": base64.b64encode(vb).decode(),
                         "bpe_b64": base64.b64encode(bb).decode(), "raises": None})
        except Exception as e:  # noqa: BLE001
            rows.append({"name": name, "vocab_b64": base64.b64encode(vb).decode(),
                         "bpe_b64": base64.b64encode(bb).decode(), "raises": type(e).__name__})
    jl("g4_loader.jsonl", rows)


# ------------------------------------------------------------------ G5
def _hash_corpus(tok, text, offs, max_len, lo, hi):
    h_ids, h_mask, ntok = hashlib.sha256(), hashlib.sha256(), 0
    raw = text.tobytes()
    for i in range(lo, hi):
        r = tok(raw[offs[i]:offs[i + 1]].decode("utf-8"), max_len=max_len)
        ids = np.asarray(r["input_ids"], dtype="<i4")
        m = np.asarray(r["attention_mask"], dtype="<i4")
        h_ids.update(ids.tobytes()); h_mask.update(m.tobytes()); ntok += int(m.sum())
    return h_ids.hexdigest(), h_mask.hexdigest(), ntok


_CACHE = {}


def _hash_worker(job):
    cfg, n, lo, hi, custom = job
    if (cfg, n) not in _CACHE:
        _CACHE.clear()
        _CACHE[(cfg, n)] = corpus.config_corpus(cfg, n_docs=n)
    text, offs, L = _CACHE[(cfg, n)]
    if custom:
        tmp = tempfile.mkdtemp()
        v, b = corpus.custom_tables()
        open(tmp + "/v", "wb").write(v); open(tmp + "/b", "wb").write(b)
        tok = Tokenize.fromFile(tmp + "/v", tmp + "/b")
    else:
        tok = Tokenize()
    return _hash_corpus(tok, text, offs, L, lo, hi)


def g5(full_cfg3: bool):
    """Per-block hashes: documents are hashed in blocks of `blk` so the work can be
    spread over processes; a checker recomputes the same per-block digests."""
    import multiprocessing as mp
    out = {}
    plans = [("cfg2_10k", 2, 10_000, 2_500, False), ("cfg3_20k", 3, 20_000, 2_500, False),
             ("cfg5_300", 5, 300, 75, True)]
    if full_cfg3:
        plans.append(("cfg3_1M", 3, 1_000_000, 25_000, False))
    with mp.Pool(8) as pool:
        for name, cfg, n, blk, custom in plans:
            jobs = [(cfg, n, lo, min(n, lo + blk), custom) for lo in range(0, n, blk)]
            res = pool.map(_hash_worker, jobs, chunksize=1)
            _, _, L = corpus.config_corpus(cfg, n_docs=8)
            out[name] = {"cfg": cfg, "n_docs": n, "max_len": L, "block": blk, "custom_tables": custom,
                         "ids_sha256": [r[0] for r in res], "mask_sha256": [r[1] for r in res],
                         "n_tokens": int(sum(r[2] for r in res))}
            print(name, out[name]["n_tokens"])
    path = os.path.join(HERE, "g5_hashes.json")
    if not full_cfg3 and os.path.exists(path):
        old = json.load(open(path))
        if "cfg3_1M" in old:
            out["cfg3_1M"] = old["cfg3_1M"]
    json.dump(out, open(path, "w"), indent=1)


# ------------------------------------------------------------------ G6
def g6(tok):
    r = random.Random(606)
    n = tok.vocab_size()
    cont = [i for w, i in tok.encoder.items() if w.endswith("@@")]
    rows = []

    def rand_ids():
        k = r.choice([0, 1, 2, 3, 5, 8, 13, 40, 100])
        out = []
        for _ in range(k):
            m = r.random()
            if m < 0.45: out.append(r.randrange(n))
            elif m < 0.75: out.append(r.choice(cont))
            elif m < 0.85: out.append(r.randrange(5))
            elif m < 0.92: out.append(r.choice([-1, -7, n, n + 1, n + 12345, 2 ** 31 - 1, -2 ** 31, 2 ** 40]))
            else: out.append(r.choice(cont))
        return out
    bundled = [rand_ids() for _ in range(400)]
    bundled += [[1, 770, 2], [], [cont[0]], [cont[0], cont[1]], [cont[0], cont[1], 5], [n], [-1]]
    rows.append({"name": "bundled", "unk_token": "<unk>", "ids": bundled, "result": [tok.decode(x) for x in bundled]})
    tok2 = Tokenize(unk_token="[không rõ]@@")
    rows.append({"name": "bundled_custom_unk", "unk_token": "[không rõ]@@", "ids": bundled[:120],
                 "result": [tok2.decode(x) for x in bundled[:120]]})
    tables = {
        "spaces_inside": "a@@ b 1\nx 1\n@@ 1\n@@@ 1\n@ 1\n\nq@@ @@ 3\n@@  z 1\ny@@ 1\n",
        "dup_ids": "a 1\nb 1\na 2\nc@@ 1\nd 1\nc@@ 9\ne 1\n",
        "special_in_vocab": "<unk> 1\na@@ 1\n</s> 1\nb 1\n<pad> 5\n",
    }
    tmp = tempfile.mkdtemp()
    for name, v in tables.items():
        vb, bb = v.encode("utf-8"), b"#v\n"
        vp, bp = os.path.join(tmp, name + ".vocab"), os.path.join(tmp, name + ".bpe")
        open(vp, "wb").write(vb); open(bp, "wb").write(bb)
        t = Tokenize.fromFile(vp, bp)
        m = t.vocab_size()
        lists = [[r.randrange(-1, m + 2) for _ in range(r.choice([0, 1, 2, 3, 4, 6, 9, 17]))] for _ in range(150)]
        lists += [list(range(m)), list(range(m - 1, -1, -1)), [i for i in range(m) for _ in range(2)]]
        rows.append({"name": name, "vocab_b64": base64.b64encode(vb).decode(), "bpe_b64": base64.b64encode(bb).decode(),
                     "unk_token": "<unk>", "decoder": sorted(t.decoder.items()), "ids": lists,
                     "result": [t.decode(x) for x in lists]})
    jl("g6_decode.jsonl", rows)


# ------------------------------------------------------------------ G7
def g7():
    from genz_tokenize import preprocess as P
    fns = {"html": P.remove_html, "unicode": P.convert_unicode, "punct": P.remove_punctuations,
           "emoji": P.remove_emoji, "url": P.remove_URL}
    r = random.Random(707)
    ws = [chr(c) for c in (0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x1C, 0x1D, 0x1E, 0x1F, 0x20, 0x85, 0xA0, 0x1680, 0x2000, 0x2003,
                           0x200A, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000)]
    edge_cps = [0x24C1, 0x24C2, 0x2500, 0x2BEF, 0x2BF0, 0x2B55, 0x2B56, 0x2640, 0x2642, 0x2643, 0x2600, 0x25FF, 0x2702, 0x27B0,
                0x200B, 0x200C, 0x200D, 0x200E, 0x23CF, 0x23E9, 0x231A, 0x231B, 0xFE0E, 0xFE0F, 0x3030, 0x3031, 0xFFFD, 0xFFFF,
                0x10000, 0x1F251, 0x1F252, 0x1F600, 0x1F64F, 0x1F926, 0x1F937, 0x10FFFF, 0x4E2D, 0xAC00, 0xD7FF, 0xE000,
                0xD800, 0xDFFF, 0x24B6, 0x1EA0, 0x0300, 0x0301, 0x0303, 0x0309, 0x0323, 0x0302, 0x031B]
    import unicodedata
    vi = ["sinh_vi\u00ean", "c\u00f4ng_ngh\u1ec7", "Vi\u1ec7t", "Nam", "\u0111\u1eb9p", "qu\u00e1", "ng\u01b0\u1eddi", "tr\u01b0\u1eddng", "h\u1ecdc",
          "\u1ee8ng_d\u1ee5ng", "Y\u1ebfn", "\u00dd", "ngh\u0129a", "kh\u00f4ng", "\u0111\u01b0\u1ee3c"]
    vi = [w.encode().decode("unicode_escape") if "\\u" in w else w for w in vi]
    pieces_html = ["<b>", "</b>", "<a href=\"http://x.vn/a?b=1\">", "<br/>", "<", ">", "<<", ">>", "<>", "< >", "<p\nclass='x'>", "<!-- c -->",
                   "a<b", "a>b", "<\u0111>"]
    pieces_url = ["http", "https", "http:", "http://a.b/c", "https://vnexpress.net/tin-tuc?x=1&y=2", "xhttp://q", "httphttp", "htt", "ttp",
                  "HTTP://x", "http\u00a0x", "http\u3000", "http\u200bx", "http\n", "(http://x)", "http\u0300"]

    def rand_text():
        k = r.choice([0, 1, 2, 3, 5, 8, 13, 21, 40, 90])
        out = []
        for _ in range(k):
            m = r.random()
            if m < 0.25:
                w = r.choice(vi)
                out.append(unicodedata.normalize("NFD", w) if r.random() < 0.5 else w)
            elif m < 0.35: out.append(r.choice(pieces_html))
            elif m < 0.45: out.append(r.choice(pieces_url))
            elif m < 0.55: out.append(chr(r.choice(edge_cps)))
            elif m < 0.62: out.append(r.choice("aâăeêioôơuưyAÂĂEÊIOÔƠUƯYbdx") + chr(r.choice([0x300, 0x301, 0x303, 0x309, 0x323, 0x302, 0x306])))
            elif m < 0.70: out.append("".join(r.choice("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~") for _ in range(r.randint(1, 3))))
            elif m < 0.76: out.append(chr(r.choice([0x1F600, 0x1F603, 0x1F44D, 0x2764, 0xFE0F, 0x1F1FB, 0x1F1F3, 0x263A, 0x2B50])))
            else: out.append("".join(r.choice("abcdehptnx01 ") for _ in range(r.randint(1, 6))))
            if r.random() < 0.6:
                out.append(r.choice(ws) if r.random() < 0.35 else " ")
        return "".join(out)

    fixed = ["", " ", "<", ">", "<>", "a<b>c", "<a><b>", "<a<b>c>d", "x<y", "x>y<", "<<<>>>", "<a\n\n>b", "a <b c", "<b>bold</b> text <i",
             "http", "http ", " http", "httpx", "http://", "see http://a.b now", "http://a http://b", "ahttphttpb c", "https", "xhttp",
             "http\u00a0", "a\u0300", "A\u0323", "a\u0302\u0301", "\u00e2\u0301", "y\u0303Y\u0309", "\u0300a", "aa\u0300\u0300",
             "o\u031b\u0301", "\u01a1\u0301", "!!!", "a.b,c", "\U0001F600", "a\U0001F600b", "a \U0001F600 b", "  a  b  ", "\u3000a\u3000b",
             "a\u200db", "\u24c1\u24c2", "x\ufe0f", "\u4e2d\u6587 text", "a\tb\nc\r\nd", "\u2028x\u2029", "tab\x1cfs", "\ud800x", "end\udfff"]
    texts = fixed + [rand_text() for _ in range(1200)]
    rows = []
    for t in texts:
        for name, fn in fns.items():
            rows.append({"ops": [name], "text": t, "result": fn(t)})
    chains = [["html", "url", "emoji"], ["unicode", "punct"], ["url", "html"], ["emoji", "punct", "unicode"],
              ["html", "unicode", "punct", "emoji", "url"], ["punct", "html"], ["emoji", "emoji"], ["url", "url"]]
    for t in texts[::3]:
        for ch in chains:
            x = t
            for name in ch:
                x = fns[name](x)
            rows.append({"ops": ch, "text": t, "result": x})
    jl("g7_preprocess.jsonl.gz", rows)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g3", "g4", "g5", "g6", "g7"]
    tok = Tokenize()
    if "g1" in which: g1(tok)
    if "g3" in which: g3(tok)
    if "g4" in which: g4()
    if "g5" in which: g5(False)
    if "g5full" in which: g5(True)
    if "g6" in which: g6(tok)
    if "g7" in which: g7()
