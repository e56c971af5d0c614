"""ctypes face of oracle/gz_oracle.c (TEST INFRASTRUCTURE ONLY -- see the header of that file).

`COracle.call_batch` returns the same ragged lists as `gz_oracle.call_batch`, so tests can compare the two
restatements, the golden vectors and the HIP path with one helper.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libgz_oracle.so")
DEFAULT_SPECIALS = ("<pad>", "<s>", "</s>", "<mask>", "<unk>")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    src = os.path.join(_HERE, "gz_oracle.c")
    if not os.path.exists(_LIB) or (os.path.exists(src) and os.path.getmtime(_LIB) < os.path.getmtime(src)):
        build()
    lib = C.CDLL(_LIB)
    lib.gzo_create.restype = C.c_void_p
    lib.gzo_create.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
    lib.gzo_destroy.argtypes = [C.c_void_p]
    lib.gzo_vocab_size.argtypes = [C.c_void_p]
    lib.gzo_n_ranks.argtypes = [C.c_void_p]
    lib.gzo_special_ids.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.gzo_lookup.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.gzo_call_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                   C.c_uint32, C.c_int64] + [C.c_void_p] * 7
    return lib


def _pack(texts):
    enc = [t.encode("utf-8", "surrogatepass") for t in texts]
    off = np.zeros(len(enc) + 1, np.int64)
    np.cumsum([len(e) for e in enc], out=off[1:])
    return np.frombuffer(b"".join(enc) + b"\0", np.uint8), off


class COracle:
    def __init__(self, vocab: bytes, merges: bytes, specials=DEFAULT_SPECIALS):
        self._lib = _load()
        arr = (C.c_char_p * 5)(*[s.encode("utf-8") for s in specials])
        err = C.c_int(0)
        self._h = self._lib.gzo_create(vocab, len(vocab), merges, len(merges), arr, C.byref(err))
        if not self._h:
            raise UnicodeDecodeError("utf-8", b"", 0, 1, "invalid table file")
        ids = (C.c_int32 * 4)()
        self._lib.gzo_special_ids(self._h, ids)
        self.pad_id, self.bos_id, self.eos_id, self.unk_id = list(ids)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.gzo_destroy(self._h)
            self._h = None

    @property
    def vocab_size(self):
        return self._lib.gzo_vocab_size(self._h)

    @property
    def n_ranks(self):
        return self._lib.gzo_n_ranks(self._h)

    def lookup(self, token: str):
        b = token.encode("utf-8", "surrogatepass")
        r = self._lib.gzo_lookup(self._h, b, len(b))
        return None if r < 0 else r

    def call_packed(self, text, toff, pair=None, poff=None, max_len=None, padding=True, truncation=True):
        """Packed in, packed out: (ids, mask, tt, seq, row_off, pair_len, status) numpy arrays."""
        n = len(toff) - 1
        flags = (1 if padding else 0) | (2 if truncation else 0) | (4 if max_len is None else 0)
        ml = 0 if max_len is None else int(max_len)
        nbytes = int(toff[-1] - toff[0]) + (int(poff[-1] - poff[0]) if pair is not None else 0)
        cap = nbytes + (4 + max(ml, 0)) * n + 16
        ids = np.empty(cap, np.int32); mask = np.empty(cap, np.int32)
        tt = np.empty(cap if pair is not None else 1, np.int32); sq = np.empty(cap if pair is not None else 1, np.int32)
        row = np.zeros(n + 1, np.int64); pl = np.zeros(2 * n, np.int32); st = np.zeros(n, np.int32)
        p = lambda a: None if a is None else a.ctypes.data
        rc = self._lib.gzo_call_batch(self._h, p(text), p(toff), p(pair), p(poff), n, ml, flags, cap,
                                      p(ids), p(mask), p(tt), p(sq), p(row), p(pl), p(st))
        assert rc == 0, rc
        return ids, mask, tt, sq, row, pl, st

    def call_batch(self, texts, pairs=None, max_len=None, padding=True, truncation=True):
        text, toff = _pack(texts)
        pair, poff = _pack(pairs) if pairs is not None else (None, None)
        ids, mask, tt, sq, row, pl, st = self.call_packed(text, toff, pair, poff, max_len, padding, truncation)
        I, M, T, S = [], [], [], []
        for d in range(len(texts)):
            a, b = int(row[d]), int(row[d + 1])
            I.append(ids[a:b].tolist()); M.append(mask[a:b].tolist())
            if pairs is None or st[d]:
                T.append(None); S.append(None)
            else:
                S.append(sq[a:a + pl[2 * d]].tolist()); T.append(tt[a:a + pl[2 * d + 1]].tolist())
        return I, M, T, S, st.tolist()
