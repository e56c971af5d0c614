#!/usr/bin/env python3
"""Per-shard digests of BASELINE configs[3] (SURVEY.md 8(d) cfg 4): 10 M documents = 8 shards x 1.25 M, shard s drawn
with seed 100 + s by corpus.config_corpus(4, ...).  Writes `cfg4_shard{0..7}` into g5_hashes.json:

  * `ids_sha256` / `mask_sha256` / `n_tokens` over the WHOLE shard, in blocks of 25 000 documents -- computed by the
    C restatement oracle/gz_oracle.c (itself pinned to the reference by tests/test_oracle_c.py on every block of
    cfg 2 / cfg 3 incl. the full 1 M documents; the REFERENCE needs ~11 min per shard per core, the C oracle 40 s);
  * `ref_prefix`: the same digests over the first 20 000 documents of the shard in blocks of 2 500 -- computed by the
    REFERENCE itself (imported from /root/reference, build container only).  tests/test_oracle_c.py checks that the
    C oracle reproduces these too, so the two sources overlap on every shard.

Also `cfg5_50k` (BASELINE configs[4] at full size), see cfg5_job.

Run only in the build container:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_shard_digests.py [all|shards|cfg5] [procs]
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

N_SHARDS, SHARD_DOCS, SEED0 = 8, 1_250_000, 100
BLOCK, REF_DOCS, REF_BLOCK = 25_000, 20_000, 2_500


def shard_job(s):
    import numpy as np
    import corpus
    import gz_oracle_c as OC
    text, offs, L = corpus.config_corpus(4, n_docs=SHARD_DOCS, seed=SEED0 + s)
    text = np.ascontiguousarray(text)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    # ---- the reference on the prefix
    sys.path.insert(0, "/root/reference")
    from genz_tokenize import Tokenize          # the reference package (this file's sys.path puts it first)
    tok = Tokenize()
    raw = text[:int(offs[REF_DOCS])].tobytes()
    r_ids, r_mask, r_tok = [], [], 0
    for lo in range(0, REF_DOCS, REF_BLOCK):
        hi_, hm_ = hashlib.sha256(), hashlib.sha256()
        for i in range(lo, lo + REF_BLOCK):
            r = tok(raw[offs[i]:offs[i + 1]].decode("utf-8"), max_len=L)
            a = np.asarray(r["input_ids"], dtype="<i4"); m = np.asarray(r["attention_mask"], dtype="<i4")
            hi_.update(a.tobytes()); hm_.update(m.tobytes()); r_tok += int(m.sum())
        r_ids.append(hi_.hexdigest()); r_mask.append(hm_.hexdigest())
    # ---- the C oracle on the whole shard
    co = OC.COracle(open(corpus.VOCAB_PATH, "rb").read(), open(corpus.BPE_PATH, "rb").read())
    c_ids, c_mask, c_tok = [], [], 0
    for lo in range(0, SHARD_DOCS, BLOCK):
        o = offs[lo:min(lo + BLOCK, SHARD_DOCS) + 1]
        ids, mask, _, _, row, _, _ = co.call_packed(text, o, max_len=L)
        k = int(row[-1])
        c_ids.append(hashlib.sha256(ids[:k].astype("<i4").tobytes()).hexdigest())
        c_mask.append(hashlib.sha256(mask[:k].astype("<i4").tobytes()).hexdigest())
        c_tok += int(mask[:k].sum())
    return s, {"cfg": 4, "shard": s, "seed": SEED0 + s, "n_docs": SHARD_DOCS, "max_len": L, "block": BLOCK,
               "custom_tables": False, "input_bytes": int(offs[-1]),
               "source": "oracle/gz_oracle.c (pinned to the reference by tests/test_oracle_c.py)",
               "ids_sha256": c_ids, "mask_sha256": c_mask, "n_tokens": c_tok,
               "ref_prefix": {"source": "reference (/root/reference genz_tokenize.Tokenize)", "n_docs": REF_DOCS,
                              "block": REF_BLOCK, "ids_sha256": r_ids, "mask_sha256": r_mask, "n_tokens": r_tok}}


def cfg5_job(_):
    """BASELINE configs[4] (SURVEY.md 8(d) cfg 5) at FULL size: 50 000 documents of <= 4 000 characters on the custom
    100 k-entry vocabulary / header-less merges (corpus.custom_tables), max_len=1024 pad+trunc AND unpadded
    (max_len=None).  Whole corpus by the C oracle, the first 300 documents also by the reference (Tokenize.fromFile)."""
    import tempfile
    import numpy as np
    import corpus
    import gz_oracle_c as OC
    n, blk, nref, rblk = 50_000, 2_500, 300, 75
    text, offs, L = corpus.config_corpus(5, n_docs=n)
    text = np.ascontiguousarray(text); offs = np.ascontiguousarray(offs, dtype=np.int64)
    v, b = corpus.custom_tables()
    sys.path.insert(0, "/root/reference")
    from genz_tokenize import Tokenize
    tmp = tempfile.mkdtemp()
    open(tmp + "/v", "wb").write(v); open(tmp + "/b", "wb").write(b)
    tok = Tokenize.fromFile(tmp + "/v", tmp + "/b")
    co = OC.COracle(v, b)
    raw = text[:int(offs[nref])].tobytes()
    out = {"cfg": 5, "n_docs": n, "max_len": L, "block": blk, "custom_tables": True, "input_bytes": int(offs[-1]),
           "source": "oracle/gz_oracle.c (pinned to the reference by tests/test_oracle_c.py)"}
    for name, ml in (("padded", L), ("unpadded", None)):
        r_ids, r_mask, r_tok = [], [], 0
        for lo in range(0, nref, rblk):
            hi_, hm_ = hashlib.sha256(), hashlib.sha256()
            for i in range(lo, lo + rblk):
                r = tok(raw[offs[i]:offs[i + 1]].decode("utf-8"), max_len=ml)
                a = np.asarray(r["input_ids"], dtype="<i4"); m = np.asarray(r["attention_mask"], dtype="<i4")
                hi_.update(a.tobytes()); hm_.update(m.tobytes()); r_tok += int(m.sum())
            r_ids.append(hi_.hexdigest()); r_mask.append(hm_.hexdigest())
        c_ids, c_mask, c_tok = [], [], 0
        for lo in range(0, n, blk):
            o = offs[lo:min(lo + blk, n) + 1]
            ids, mask, _, _, row, _, _ = co.call_packed(text, o, max_len=ml)
            k = int(row[-1])
            c_ids.append(hashlib.sha256(ids[:k].astype("<i4").tobytes()).hexdigest())
            c_mask.append(hashlib.sha256(mask[:k].astype("<i4").tobytes()).hexdigest())
            c_tok += int(mask[:k].sum())
        out[name] = {"max_len": ml, "ids_sha256": c_ids, "mask_sha256": c_mask, "n_tokens": c_tok,
                     "ref_prefix": {"source": "reference (Tokenize.fromFile)", "n_docs": nref, "block": rblk,
                                    "ids_sha256": r_ids, "mask_sha256": r_mask, "n_tokens": r_tok}}
    return out


if __name__ == "__main__":
    import multiprocessing as mp
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    path = os.path.join(HERE, "g5_hashes.json")
    out = json.load(open(path))
    with mp.get_context("spawn").Pool(procs) as pool:
        if what in ("all", "cfg5"):
            out["cfg5_50k"] = pool.map(cfg5_job, [0])[0]
            print("cfg5_50k", out["cfg5_50k"]["padded"]["n_tokens"], out["cfg5_50k"]["unpadded"]["n_tokens"])
        if what in ("all", "shards"):
            res = dict(pool.map(shard_job, range(N_SHARDS), chunksize=1))
            for s in range(N_SHARDS):
                out["cfg4_shard%d" % s] = res[s]
                print(s, res[s]["n_tokens"], res[s]["input_bytes"], res[s]["ref_prefix"]["n_tokens"])
    json.dump(out, open(path, "w"), indent=1)
