"""CPU-only: the C-ABI library loads without a GPU and exports every symbol include/genz_tokenize.h declares;
no compute entry point is called here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "genz_tokenize.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gz_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    native = pytest.importorskip("genz_tokenize._native")
    lib = native.load_library()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert sorted(native.SYMBOLS) == names          # the ctypes binding covers the whole header
    assert lib.gz_version() == 0x010100


def test_limits_of_the_32_bit_paths():
    """gz_limit: one text of an encode call is refused from 0xFFFF0000 bytes on (32-bit positions on the device); the text
    pre-pass switches from its 32-bit length scan to the 64-bit one at an INPUT size below 2^32 (filters never grow a document, so
    under that size no output offset and no total can wrap) -- round-4 advice: the 32-bit tail used to be taken at any size."""
    native = pytest.importorskip("genz_tokenize._native")
    lib = native.load_library()
    enc, pp = lib.gz_limit(0), lib.gz_limit(1)
    assert 0 < enc < 2**32 and 0 < pp < 2**32
    assert pp <= 2**32 - 65536                      # slack for the kernels' 16-byte accesses behind the last document
    assert lib.gz_limit(2) == -1 and lib.gz_limit(-1) == -1


def test_retired_host_table_answers_invalid_and_clears_its_outputs():
    native = pytest.importorskip("genz_tokenize._native")
    ht = native.HostTables(b"a 1\nb 1\n", b"#version: 0.2\na b\n")
    p, n = ctypes.c_void_p(123), ctypes.c_int64(456)
    for which in (0, 10, -1):
        p.value, n.value = 123, 456
        assert ht.lib.gz_host_tables_array(ht.handle, which, ctypes.byref(p), ctypes.byref(n)) == native.GZ_E_INVALID
        assert not p.value and n.value == 0
    assert len(ht.array(1)) == 2
    ht.close()


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a usable device the drop-in must raise, never compute on the CPU."""
    native = pytest.importorskip("genz_tokenize._native")
    lib = native.load_library()
    h = ctypes.c_void_p()
    rc = lib.gz_create(0, ctypes.byref(h))
    if rc == 0:                                     # a GPU is present (the GPU box): nothing to check here
        lib.gz_destroy(h)
        pytest.skip("a HIP device is available")
    assert rc == native.GZ_E_NODEVICE
    assert b"no HIP device" in lib.gz_last_error(None) or b"gfx950" in lib.gz_last_error(None)
    from genz_tokenize import Tokenize
    with pytest.raises(RuntimeError):
        Tokenize()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "genz-tokenize_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".inc")):
                text = open(os.path.join(dirpath, f), encoding="utf-8", errors="replace").read()
                assert "gz_oracle" not in text and "oracle/" not in text, os.path.join(dirpath, f)


def test_string_packer_matches_python_encoding():
    """csrc/gz_pack.c (host-side packing of list[str] for the batch calls) against `str.encode("utf-8", "surrogatepass")`:
    ASCII, Latin-1, BMP, astral, lone surrogates, empty strings, tuples; a non-str item raises the reference's TypeError."""
    import random
    import numpy as np
    _packing = pytest.importorskip("genz_tokenize._packing")
    if _packing._gz_pack is None:
        pytest.skip("_gz_pack is not built (make -C genz-tokenize_amd/csrc pack)")
    r = random.Random(5)
    alphabet = ["a", " ", "z", "\x7f", "\x80", "\xff", "Ā", "߿", "ࠀ", "ấ", "￿", "\ud800", "\udfff",
                "\U00010000", "\U0001F600", "\U0010FFFF", "\n", "　"]
    docs = ["".join(r.choice(alphabet) for _ in range(r.choice([0, 1, 2, 5, 17, 200]))) for _ in range(3000)]
    for seq in (docs, tuple(docs[:50]), [], ["only ascii"] * 9, ["\xe9"] * 9):
        b, o = _packing._gz_pack.pack(seq)
        parts = [d.encode("utf-8", "surrogatepass") for d in seq]
        assert b == b"".join(parts)
        assert np.frombuffer(o, np.int64).tolist() == [0] + np.cumsum([len(p) for p in parts], dtype=np.int64).tolist()
    # pack_into: the same bytes into the caller's buffers (a pinned arena in the product); too small -> -(bytes needed), nothing written
    want = b"".join(d.encode("utf-8", "surrogatepass") for d in docs)
    arena = np.full(len(want) + 100, 0xEE, dtype=np.uint8)
    offs = np.empty(len(docs) + 1, dtype=np.int64)
    assert _packing._gz_pack.pack_into(docs, arena, offs) == len(want)
    assert arena[:len(want)].tobytes() == want and (arena[len(want):] == 0xEE).all()
    assert offs.tolist() == [0] + np.cumsum([len(d.encode("utf-8", "surrogatepass")) for d in docs], dtype=np.int64).tolist()
    small = np.full(len(want) - 1, 0xEE, dtype=np.uint8)
    assert _packing._gz_pack.pack_into(docs, small, offs) == -len(want) and (small == 0xEE).all()
    with pytest.raises(ValueError):
        _packing._gz_pack.pack_into(docs, arena, np.empty(len(docs), dtype=np.int64))
    with pytest.raises(TypeError, match="expected string or bytes-like object"):
        _packing._gz_pack.pack_into(docs[:8] + [None], arena, offs)
    buf, off = _packing.pack(docs)
    assert buf.tobytes() == b"".join(d.encode("utf-8", "surrogatepass") for d in docs)
    assert off.dtype == np.int64 and len(off) == len(docs) + 1
    for bad in (["a"] * 8 + [3], ["a"] * 8 + [b"x"], ["a"] * 8 + [None]):
        with pytest.raises(TypeError, match="expected string or bytes-like object"):
            _packing.pack(bad)


def test_host_expand_of_csr_rows_matches_numpy():
    """csrc/gz_pack.c expand(): CSR rows -> dense [N, L] input_ids (padded with the pad id) + attention_mask = ids != pad,
    on threads; against a plain numpy restatement (16- and 32-bit entries, empty rows, full rows, a real token equal to the pad id)."""
    import numpy as np
    _packing = pytest.importorskip("genz_tokenize._packing")
    if _packing._gz_pack is None or not hasattr(_packing._gz_pack, "expand"):
        pytest.skip("_gz_pack is not built (make -C genz-tokenize_amd/csrc pack)")
    rng = np.random.default_rng(4)
    for n, L, pad, bits in ((1, 4, 0, 16), (1000, 16, 0, 16), (30000, 32, 7, 32), (25000, 8, 3, 16)):
        n_real = rng.integers(0, L + 1, size=n).astype(np.int32)
        row = np.zeros(n + 1, dtype=np.int64); np.cumsum(n_real, out=row[1:])
        toks = rng.integers(0, 60000, size=int(row[-1])).astype(np.uint16 if bits == 16 else np.int32)
        if len(toks) > 5:
            toks[::5] = pad                                                  # real tokens equal to the pad id: mask 0 there
        ids = np.empty((n, L), dtype=np.int32); mask = np.empty((n, L), dtype=np.int32)
        _packing._gz_pack.expand(toks, bits, n_real, row, L, pad, ids, mask)
        want = np.full((n, L), pad, dtype=np.int32)
        sel = np.arange(L)[None, :] < n_real[:, None]
        want[sel] = toks.astype(np.int32)
        assert np.array_equal(ids, want) and np.array_equal(mask, (want != pad).astype(np.int32))
    with pytest.raises(ValueError):
        _packing._gz_pack.expand(np.zeros(3, np.uint16), 16, np.array([2, 2], np.int32), np.array([0, 2, 4], np.int64), 4, 0,
                                 np.empty((2, 4), np.int32), np.empty((2, 4), np.int32))       # fewer entries than the rows announce


def test_batch_offset_spans_from_word_counts():
    """Host logic of `encode_batch(..., return_offset=True)`: the numpy span arithmetic over per-word piece counts against the
    list form of tokenize.py:105, :111-117 (and :231-234 for pairs: B's entries shifted by A's entry count), with a fake
    context that serves random counts -- documents without words included."""
    import numpy as np
    from genz_tokenize.tokenize import Tokenize

    class Ctx:
        def __init__(self, per_text):
            self.t = per_text

        def word_token_counts(self, which, n, cap):
            docs = self.t[which]
            counts = np.array([c for d in docs for c in d], dtype=np.int32)
            first = np.zeros(n + 1, np.int64)
            np.cumsum([len(d) for d in docs], out=first[1:])
            return counts, first

    def ref(doc):
        off, seen = [(0, 0)], 0
        for c in doc:
            off.append((seen + 1, seen + c)); seen += c
        off.append((seen + 1, seen + 1))
        return off

    rng = np.random.default_rng(1)
    n = 64
    A = [list(map(int, rng.integers(1, 5, size=rng.integers(0, 6)))) for _ in range(n)]
    B = [list(map(int, rng.integers(1, 5, size=rng.integers(0, 6)))) for _ in range(n)]
    t = Tokenize.__new__(Tokenize)
    t._ctx = Ctx([A, B])
    o, e = t._batch_offsets(n, False)
    r = dict(offset=o, offset_off=e)
    assert o.dtype == np.int32 and all(Tokenize.offsets_of(r, i) == ref(A[i]) for i in range(n))
    o, e = t._batch_offsets(n, True)
    r = dict(offset=o, offset_off=e)
    for i in range(n):
        ra = ref(A[i])
        assert Tokenize.offsets_of(r, i) == ra + [(a + len(ra), b + len(ra)) for a, b in ref(B[i])], i
