"""Drop-in `Tokenize` for DVNghiem/genz-tokenize v1.2.7, computed on an MI355X.

Same Python surface as the reference class (reference genz_tokenize/tokenize.py:6-278):
`Tokenize(...)`, `Tokenize.fromFile`, `__call__`, `encode`, `decode`, `bpe`, `vocab_size`,
`add_vocab_file`, `add_bpe_file`, the three `get_*` helpers and the module function
`get_pairs`, with the same results, key order, `None`s and exceptions.  Underneath, every
call that tokenizes goes through the C ABI of include/genz_tokenize.h into hand-written HIP
kernels; nothing here falls back to a CPU implementation.

On top of the reference surface there is a batch API (`encode_batch`, `encode_packed`)
that returns numpy arrays, which is what the GPU is for.
"""
from __future__ import annotations

import os
import threading
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _native, _packing
from ._packing import pack as _pack            # list of str -> packed UTF-8 + offsets (C when built)

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def _read_text(path) -> str:
    # the reference opens both files with open(path, 'r', encoding='utf-8') (tokenize.py:45, :54):
    # FileNotFoundError / UnicodeDecodeError surface from here exactly as they do there
    with open(path, "r", encoding="utf-8") as f:
        return f.read()


def _require_str(x):
    # re.findall(r"\S+\n?", x) (tokenize.py:106) is what rejects non-str input in the reference
    if isinstance(x, str):
        return
    if isinstance(x, (bytes, bytearray, memoryview)):
        raise TypeError("cannot use a string pattern on a bytes-like object")
    raise TypeError("expected string or bytes-like object")


class Tokenize(object):
    def __init__(self, pad_token='<pad>', bos_token='<s>', eos_token='</s>', mask_token='<mask>',
                 unk_token='<unk>', device: Optional[int] = None) -> None:
        super().__init__()
        # tokenize.py:15-23 -- paths survive a second __init__ (that is how fromFile works)
        if not hasattr(self, "vocab_file"):
            self.vocab_file = os.path.join(_DATA, "vocab.txt")
        if not hasattr(self, "bpe_file"):
            self.bpe_file = os.path.join(_DATA, "bpe.codes")
        self.pad_token, self.bos_token, self.eos_token = pad_token, bos_token, eos_token
        self.mask_token, self.unk_token = mask_token, unk_token
        if not hasattr(self, "_ctx"):
            self._ctx = _native.Context(device)
            self._batch_lock = threading.Lock()      # the batch calls' pinned text arena (one per object): _packing.pack_pinned
        self._vocab_texts: List[str] = []
        self._bpe_text = ""
        self._dicts = {}
        self._dirty = True
        self._snapshot_pending = False
        self.add_vocab_file(self.vocab_file)
        # tokenize.py:40: `decoder` is built ONCE, here, from the first vocab load; a later add_vocab_file does not
        # change it.  Both snapshots are taken lazily: the host dict when `.decoder` / decode() is first used, the
        # device one with the first table build (the merges loaded next do not touch `encoder`).
        self._decoder_snapshot = None
        self._snapshot_source = (self._vocab_texts[0], (pad_token, bos_token, eos_token, mask_token, unk_token))
        self._snapshot_pending = True
        self.add_bpe_file(self.bpe_file)

    # ---- tables -------------------------------------------------------------------------------------------
    def add_vocab_file(self, vocab_file):
        """tokenize.py:44-51 -- appends words to `encoder`; the device tables are rebuilt lazily."""
        text = _read_text(vocab_file)
        if getattr(self, "_snapshot_pending", False):
            self._sync_tables()                 # take the device decoder snapshot before the vocabulary grows
        self._vocab_texts.append(text)
        self._dirty = True

    def add_bpe_file(self, bpe_file):
        """tokenize.py:53-57 -- replaces `bpe_ranks`."""
        self._bpe_text = _read_text(bpe_file)
        self._dirty = True

    def _sync_tables(self):
        if not self._dirty:
            return
        # several vocab files behave like one file whose lines are the concatenation of theirs
        chunks = []
        for t in self._vocab_texts:
            chunks.append(t if (t == "" or t.endswith("\n")) else t + "\n")
        vocab = "".join(chunks).encode("utf-8")
        self._ctx.load_tables(vocab, self._bpe_text.encode("utf-8"),
                              (self.pad_token, self.bos_token, self.eos_token, self.mask_token, self.unk_token))
        self._dicts = {}
        self._dirty = False
        if self._snapshot_pending:
            self._ctx.decoder_snapshot()
            self._snapshot_pending = False

    @property
    def encoder(self) -> Dict[str, int]:
        self._sync_tables()
        if "encoder" not in self._dicts:
            self._dicts["encoder"] = dict(self._ctx.vocab_items())
        return self._dicts["encoder"]

    @property
    def decoder(self) -> Dict[int, str]:
        if self._decoder_snapshot is None:
            text, specials = self._snapshot_source
            vocab = (text if (text == "" or text.endswith("\n")) else text + "\n").encode("utf-8")
            ht = _native.HostTables(vocab, b"", specials)
            self._decoder_snapshot = {v: k for k, v in ht.vocab_items()}
            ht.close()
        return self._decoder_snapshot

    @property
    def bpe_ranks(self):
        self._sync_tables()
        if "ranks" not in self._dicts:
            self._dicts["ranks"] = dict(self._ctx.merge_items())
        return self._dicts["ranks"]

    def vocab_size(self):
        self._sync_tables()
        return self._ctx.table_info()[0]

    def _special_ids(self):
        self._sync_tables()
        return self._ctx.table_info()[1]       # pad, bos, eos, mask, unk  (looked up at call time, rule L4)

    # ---- BPE of one word -----------------------------------------------------------------------------------
    def bpe(self, token):
        """tokenize.py:62-101: "piece@@ piece ..." for ONE word (no whitespace splitting here)."""
        if not isinstance(token, str):
            raise TypeError("bpe() expects str")
        if token == "":
            raise IndexError("tuple index out of range")         # word[-1] on an empty tuple (tokenize.py:64)
        self._sync_tables()
        pieces = self._ctx.bpe_word(token.encode("utf-8", "surrogatepass"))
        if len(pieces) == 1 and len(token) == 1:
            return token                                         # tokenize.py:66-67
        strs = []
        for k, p in enumerate(pieces):
            p = int(p)
            if p >= 0:
                strs.append(self._ctx.symbol(p))
            else:
                strs.append(chr(-p - 1) + ("</w>" if k == len(pieces) - 1 else ""))
        return "@@ ".join(strs)[:-4]

    # ---- encode / decode -------------------------------------------------------------------------------------
    def _run(self, texts, pairs, max_len, padding, truncation, word_table=True, keep_words=False):
        self._sync_tables()
        tb, to = _pack(texts)
        pb = po = None
        if pairs is not None:
            pb, po = _pack(pairs)
        return self._ctx.encode(tb, to, pb, po, max_len, bool(padding), bool(truncation),
                                (0 if word_table else _native.GZ_NO_WORD_TABLE) | (_native.GZ_KEEP_WORDS if keep_words else 0))

    def encode(self, sentence, return_offset) -> List[int]:
        """tokenize.py:126-135."""
        _require_str(sentence)
        r = self._run([sentence], None, None, False, False, keep_words=bool(return_offset))
        ids = r["input_ids"].tolist()
        if return_offset:
            return ids, self._offsets(0, len(sentence.encode("utf-8", "surrogatepass")))
        return ids

    def _offsets(self, which_text, nbytes):
        """tokenize.py:105, :111-117: [(0,0)] + one (first, last) 1-based token span per word + (T+1, T+1), from the
        per-word piece counts of the last device call."""
        counts, _ = self._ctx.word_token_counts(which_text, 1, nbytes // 1 + 2)
        off, seen = [(0, 0)], 0
        for c in counts.tolist():
            off.append((seen + 1, seen + c)); seen += c
        off.append((seen + 1, seen + 1))
        return off

    def decode(self, token):
        """tokenize.py:137-139 (host side: string assembly)."""
        dec, unk = self.decoder, self.unk_token
        return ' '.join(dec.get(i, unk) for i in token).replace('@@ ', '')

    def decode_batch(self, rows) -> List[str]:
        """`decode` for many id lists at once, on the GPU (the reference has no batch form).
        `rows` is a sequence of integer sequences or a 2-D integer array; ids outside the int32 range decode to the
        unk token like any other unknown id."""
        if isinstance(rows, np.ndarray) and rows.ndim == 2:
            n, L = rows.shape
            flat = rows.reshape(-1)
            off = np.arange(n + 1, dtype=np.int64) * L
        else:
            rows = [np.asarray(r, dtype=np.int64).reshape(-1) for r in rows]
            off = np.zeros(len(rows) + 1, dtype=np.int64)
            if rows:
                np.cumsum([len(r) for r in rows], out=off[1:])
            flat = np.concatenate(rows) if rows else np.zeros(0, dtype=np.int64)
        self._sync_tables()
        flat = np.asarray(flat)
        if flat.dtype != np.int32:
            if flat.dtype.kind not in "iu":
                raise TypeError("decode_batch() expects integer ids")
            flat = np.where((flat < -(2 ** 31)) | (flat >= 2 ** 31), -1, flat).astype(np.int32)
        data, out_off = self._ctx.decode(flat, off, self.unk_token.encode("utf-8", "surrogatepass"))
        raw = data.tobytes()
        return [raw[out_off[i]:out_off[i + 1]].decode("utf-8", "surrogatepass") for i in range(len(off) - 1)]

    def get_atttention_mask(self, token):
        pad = self._special_ids()[0]
        return [1 if i != pad else 0 for i in token]

    def get_token_type(self, token):
        """tokenize.py:154-161 (mutates and returns its argument; ValueError when < 2 None remain)."""
        token[0] = 0
        token[-1] = 1
        token[token.index(None)] = 0
        token[token.index(None)] = 1
        return token

    def get_sequence_id(self, token):
        """tokenize.py:163-182."""
        sp = self._special_ids()
        bos, eos = sp[1], sp[2]
        out = []
        for v in token:
            if v == eos:
                out.append(None)
                break
            out.append(None if v == bos else 0)
        for k in range(len(out), len(token)):
            if token[k] == eos:
                out.append(None)
                if out[k - 1] == 1:
                    break
            else:
                out.append(1)
        return out

    def __call__(self, text: str, pair_text: str = None, max_len: int = None, padding: bool = True,
                 truncation: bool = True, return_offset: bool = False) -> Dict:
        """tokenize.py:184-259.  Returns {'input_ids', 'attention_mask'[, 'sequence_id', 'token_type_ids']}."""
        _require_str(text)
        if pair_text is not None:
            _require_str(pair_text)
        if max_len is not None and not isinstance(max_len, (int, np.integer)):
            raise TypeError("max_len must be an int or None")
        r = self._run([text], None if pair_text is None else [pair_text], max_len, padding, truncation, keep_words=bool(return_offset))
        if pair_text is not None and int(r["status"][0]) != 0:
            raise ValueError("None is not in list")                 # tokenize.py:157-160, rule P3
        result = {}
        if return_offset:                                            # tokenize.py:225-234, :241-244 (first key, rule R1)
            off = self._offsets(0, len(text.encode("utf-8", "surrogatepass")))
            if pair_text is not None:
                shift = len(off)                                     # entries of A, not tokens (rule O2)
                off = off + [(a + shift, b + shift)
                             for a, b in self._offsets(1, len(pair_text.encode("utf-8", "surrogatepass")))]
            result['offset'] = off
        result['input_ids'] = r["input_ids"].tolist()
        result['attention_mask'] = r["attention_mask"].tolist()
        if pair_text is not None:
            ns, nt = (int(x) for x in r["pair_len"][0])
            seq = [None if v == _native.GZ_NONE else v for v in r["sequence_id"][:ns].tolist()]
            if max_len is not None and padding:
                tt = [None if v == _native.GZ_NONE else v for v in r["token_type_ids"][:nt].tolist()]
            else:
                tt = seq                                             # the same list object (tokenize.py:254-255)
            result['sequence_id'] = seq
            result['token_type_ids'] = tt
        return result

    # ---- batch API (new) -----------------------------------------------------------------------------------------
    def encode_batch(self, texts: Sequence[str], pair_texts: Optional[Sequence[str]] = None,
                     max_len: Optional[int] = None, padding: bool = True, truncation: bool = True,
                     word_table: bool = True, return_offset: bool = False):
        """`__call__` over many documents in one launch.  With max_len >= 1, padding and truncation the result
        arrays are [N, max_len] int32; otherwise they are flat with `row_off` [N+1].  `status[i] == 1` marks a
        document for which the single-call API raises ValueError.
        `return_offset=True` adds `__call__`'s 'offset' lists for every document (tokenize.py:105, :111-117, :225-234) as two
        arrays: `offset` int32 [E, 2] -- the (first, last) spans of all documents one after the other -- and `offset_off`
        int64 [N+1]: document i's list is offset[offset_off[i]:offset_off[i+1]] (`offsets_of(result, i)` gives it in the
        reference's list-of-tuples form)."""
        if pair_texts is not None and len(pair_texts) != len(texts):
            raise ValueError("texts and pair_texts differ in length")
        if (pair_texts is None and padding and truncation and max_len is not None and int(max_len) >= 1 and len(texts) >= 20000
                and _packing._gz_pack is not None and not return_offset):
            return self._shape(self._encode_batch_large(texts, int(max_len), word_table), len(texts))
        r = self._run(list(texts), None if pair_texts is None else list(pair_texts), max_len, padding, truncation,
                      word_table, keep_words=bool(return_offset))
        out = self._shape(r, len(texts))
        if return_offset:
            out["offset"], out["offset_off"] = self._batch_offsets(len(texts), pair_texts is not None)
        return out

    def _batch_offsets(self, n, is_pair):
        """The 'offset' lists of the last batch call, from the per-word piece counts the device kept (GZ_KEEP_WORDS): per text
        [(0, 0)] + one 1-based (first, last) token span per word + (T+1, T+1) (tokenize.py:105, :111-117); in pair mode B's
        entries follow A's, each shifted by the NUMBER OF ENTRIES of A (tokenize.py:231-234: `len(offset)`, not tokens)."""
        def spans(which):
            counts, first = self._ctx.word_token_counts(which, n, 1 << 16)
            first = first.astype(np.int64)
            nw = np.diff(first)
            eoff = first + 2 * np.arange(n + 1, dtype=np.int64)          # entries before document d: its words + 2 per document
            cs = np.zeros(len(counts) + 1, dtype=np.int64)
            np.cumsum(counts, out=cs[1:])
            base = cs[first[:-1]]
            out = np.zeros((int(eoff[-1]), 2), dtype=np.int64)
            doc = np.repeat(np.arange(n, dtype=np.int64), nw)
            w = np.arange(len(counts), dtype=np.int64)
            pos = w + 2 * doc + 1
            out[pos, 0] = cs[:-1] - base[doc] + 1
            out[pos, 1] = cs[1:] - base[doc]
            T = cs[first[1:]] - base
            out[eoff[1:] - 1] = (T + 1)[:, None]
            return out, eoff
        a, ea = spans(0)
        if not is_pair:
            return a.astype(np.int32), ea
        b, eb = spans(1)
        na, nb = np.diff(ea), np.diff(eb)
        b += np.repeat(na, nb)[:, None]                              # rule O2: shifted by A's entry count
        eoff = ea + eb
        out = np.empty((int(eoff[-1]), 2), dtype=np.int64)
        ia = np.arange(len(a), dtype=np.int64) + np.repeat(eb[:-1], na)          # A's entries of document d move up by B's earlier ones
        ib = np.arange(len(b), dtype=np.int64) + np.repeat(ea[1:], nb)           # B's follow A's of the same document
        out[ia] = a
        out[ib] = b
        return out.astype(np.int32), eoff

    @staticmethod
    def offsets_of(result, i):
        """Document i's 'offset' list of an `encode_batch(..., return_offset=True)` result, as `__call__` returns it."""
        lo, hi = int(result["offset_off"][i]), int(result["offset_off"][i + 1])
        return [(int(a), int(b)) for a, b in result["offset"][lo:hi]]

    def _encode_batch_large(self, texts, max_len, word_table):
        """Large dense single-text batches: the strings are packed on threads (csrc/gz_pack.c) straight into a pinned arena this
        object keeps, and the packed text goes through the library's dense host path (gz_encode_batch: sub-batches, only the text
        and the rows' real entries cross PCIe, the [N, max_len] arrays are padded into place by host threads inside the call).
        The arena is ONE buffer per object: the lock keeps a second thread's batch out of it until this call has consumed it."""
        self._sync_tables()
        with self._batch_lock:
            tb, to = _packing.pack_pinned(texts, self, self._ctx)
            return self._ctx.encode(tb, to, None, None, max_len, True, True, 0 if word_table else _native.GZ_NO_WORD_TABLE)

    def encode_packed(self, text_u8: np.ndarray, offsets: np.ndarray, pair_u8=None, pair_offsets=None,
                      max_len: Optional[int] = None, padding: bool = True, truncation: bool = True,
                      word_table: bool = True):
        """Batch call on already packed UTF-8 (uint8 array + int64 offsets[N+1]); skips Python string packing.
        `word_table=False` makes every word run the merge loop (same results; for tests and measurements)."""
        self._sync_tables()
        r = self._ctx.encode(text_u8, offsets, pair_u8, pair_offsets, max_len, bool(padding), bool(truncation),
                             0 if word_table else _native.GZ_NO_WORD_TABLE)
        return self._shape(r, len(offsets) - 1)

    def encode_packed_csr(self, text_u8: np.ndarray, offsets: np.ndarray, max_len: int, word_table: bool = True):
        """The fast host path for large batches: `__call__(text, max_len=max_len)` over packed UTF-8, with the result in
        CSR form -- dict(tokens [total] (uint16 when every id fits, else int32), n_real [N], row_off [N+1], max_len,
        pad_id).  Only the text and the rows' real entries cross PCIe (the copies overlap the kernels); `csr_to_dense`
        rebuilds the [N, max_len] input_ids / attention_mask of `encode_packed` on the host when they are wanted."""
        self._sync_tables()
        if max_len is None or int(max_len) < 1:
            raise ValueError("encode_packed_csr needs max_len >= 1 (rows are cut to max_len)")
        bits = 16 if self.vocab_size() <= 65536 and max(self._special_ids()) < 65536 and min(self._special_ids()) >= 0 else 32
        try:
            tokens, n_real = self._ctx.encode_csr(text_u8, offsets, int(max_len), bits, 0 if word_table else _native.GZ_NO_WORD_TABLE)
        except _native.GzError as e:
            if bits == 16 and e.code == _native.GZ_E_LIMIT:              # an id collision pushed an id past 65535
                tokens, n_real = self._ctx.encode_csr(text_u8, offsets, int(max_len), 32, 0 if word_table else _native.GZ_NO_WORD_TABLE)
            else:
                raise
        row_off = np.zeros(len(n_real) + 1, dtype=np.int64)
        np.cumsum(n_real, out=row_off[1:])
        return dict(tokens=tokens, n_real=n_real, row_off=row_off, max_len=int(max_len), pad_id=self._special_ids()[0])

    @staticmethod
    def csr_to_dense(csr):
        """(input_ids, attention_mask) [N, max_len] int32 of a `encode_packed_csr` result: rows padded with the pad id
        (tokenize.py:141-146), mask = ids != pad (:148-152)."""
        n, L = len(csr["n_real"]), csr["max_len"]
        ids = np.full((n, L), csr["pad_id"], dtype=np.int32)
        cols = np.arange(L, dtype=np.int64)[None, :]
        keep = cols < csr["n_real"][:, None]
        ids[keep] = csr["tokens"].astype(np.int32, copy=False)
        return ids, (ids != csr["pad_id"]).astype(np.int32)

    def encode_to_device(self, texts: Sequence[str], pair_texts: Optional[Sequence[str]] = None, max_len: int = 128):
        """Batch `__call__` (padding=True, truncation=True) whose [N, max_len] int32 outputs STAY in HBM: a dict with the field names
        of the reference's DataCollection (models/bert/dataset.py:7-28) -- input_ids, attention_mask and, with pair
        texts, token_type_ids / sequence_id -- as `handoff.DeviceArray`s (DLPack: `torch.from_dlpack(x)` is zero-copy),
        plus host arrays n_real [N] and status [N]."""
        from .handoff import DeviceArray
        if max_len is None or int(max_len) < 1:
            raise ValueError("encode_to_device needs max_len >= 1 (dense rows)")
        self._sync_tables()
        with self._batch_lock:                                         # (the pinned text arena is one buffer per object: see pack_pinned)
            return self._encode_to_device_locked(texts, pair_texts, int(max_len))

    def _encode_to_device_locked(self, texts, pair_texts, L):
        from .handoff import DeviceArray
        ctx = self._ctx
        t, to = _packing.pack_pinned(texts, self, ctx)                 # (a pinned arena this object keeps: the H2D copy is real DMA)
        n = len(to) - 1
        pair = pair_texts is not None
        if pair:
            if len(pair_texts) != n:
                raise ValueError("texts and pair_texts differ in length")
            p, po = _pack(pair_texts)
        bufs = []

        def up(a):
            d = ctx.alloc(max(a.nbytes, 1) + 64)
            bufs.append(d)
            if a.nbytes:
                ctx.h2d(d, np.ascontiguousarray(a))
            return d
        try:
            d_t, d_to = up(t), up(to)
            d_p, d_po = (up(p), up(po)) if pair else (0, 0)
            cells = max(n * L, 1)
            out = {k: DeviceArray(ctx, ctx.alloc(4 * cells), (n, L)) for k in
                   (("input_ids", "attention_mask", "token_type_ids", "sequence_id") if pair else ("input_ids", "attention_mask"))}
            d_nr, d_st, d_pl = ctx.alloc(4 * max(n, 1)), ctx.alloc(4 * max(n, 1)), ctx.alloc(8 * max(n, 1))
            bufs += [d_nr, d_st, d_pl]
            flags = _native.GZ_PADDING | _native.GZ_TRUNCATION
            ctx.encode_device(d_t, d_to, d_p, d_po, n, L, flags, n * L, out["input_ids"].ptr, out["attention_mask"].ptr,
                              out["token_type_ids"].ptr if pair else None, out["sequence_id"].ptr if pair else None,
                              None, d_pl if pair else None, d_nr, d_st if pair else None)
            ctx.sync()
            nr = np.zeros(n, dtype=np.int32); st = np.zeros(n, dtype=np.int32)
            if n:
                ctx.d2h(nr, d_nr)
                if pair:
                    ctx.d2h(st, d_st)
            out["n_real"], out["status"] = nr, st
            return out
        finally:
            for d in bufs:
                ctx.free(d)

    @staticmethod
    def _shape(r, n):
        if r["dense"]:
            L = r["max_len"]
            for k in ("input_ids", "attention_mask", "token_type_ids", "sequence_id"):
                if k in r:
                    r[k] = r[k].reshape(n, L)
        return r

    @classmethod
    def fromFile(cls, vocab_file, bpe_file):
        """tokenize.py:261-267.  (The reference loads the bundled tables first and then re-runs __init__;
        here the paths are set before the only load.)"""
        tok = cls.__new__(cls)
        tok.vocab_file = vocab_file
        tok.bpe_file = bpe_file
        tok.__init__()
        return tok


def get_pairs(word):
    """tokenize.py:270-278: the set of adjacent symbol pairs of a word (a tuple of symbols, or a str).  An empty word
    raises IndexError as the reference's `word[0]` does (:272)."""
    word[0]
    return {(a, b) for a, b in zip(word, word[1:])}
