"""ctypes binding of include/genz_tokenize.h (libgenz_tokenize_hip.so).

Thin by design: argument marshalling only.  There is no CPU fallback -- if the
shared library is missing or no gfx950 device is usable, constructing a context
raises RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (GZ_LIBRARY: another build of the same library -- the diagnostic build of `make diag` in the tests)
LIB_PATH = os.environ.get("GZ_LIBRARY") or os.path.join(_HERE, "libgenz_tokenize_hip.so")

GZ_OK, GZ_E_INVALID, GZ_E_UTF8, GZ_E_HIP, GZ_E_NOTABLES = 0, -1, -2, -3, -4
GZ_E_CAPACITY, GZ_E_LIMIT, GZ_E_NOMEM, GZ_E_RCCL, GZ_E_NODEVICE = -5, -6, -7, -8, -9
GZ_PADDING, GZ_TRUNCATION, GZ_MAX_LEN_NONE, GZ_TIMING, GZ_NO_WORD_TABLE, GZ_KEEP_WORDS = 0x1, 0x2, 0x4, 0x100, 0x200, 0x400
GZ_NONE = -1
GZ_PP_HTML, GZ_PP_UNICODE, GZ_PP_PUNCT, GZ_PP_EMOJI, GZ_PP_URL = 1, 2, 3, 4, 5

# every symbol include/genz_tokenize.h declares (tests/test_abi.py checks the export list against the header)
SYMBOLS = [
    "gz_version", "gz_create", "gz_destroy", "gz_last_error", "gz_load_tables", "gz_table_info", "gz_table_cache_status", "gz_table_digest",
    "gz_vocab_entry", "gz_merge_entry", "gz_encode_batch", "gz_encode_batch_csr", "gz_host_alloc", "gz_host_free", "gz_encode_batch_device", "gz_encode_batch_device_h", "gz_sync",
    "gz_word_token_counts", "gz_bpe_word", "gz_symbol_utf8", "gz_device_alloc", "gz_device_free", "gz_memcpy_h2d", "gz_memcpy_d2h",
    "gz_timing", "gz_timing_history", "gz_decoder_snapshot", "gz_decode_batch", "gz_decode_batch_device", "gz_preprocess_batch",
    "gz_preprocess_batch_device", "gz_block_create", "gz_block_release", "gz_block_dlpack", "gz_dlpack_capsule_destructor", "gz_exchange_select", "gz_encode_emit_block", "gz_block_total", "gz_comm_unique_id", "gz_comm_init", "gz_gather_rows", "gz_exchange_timing_history", "gz_compact_rows", "gz_expand_rows", "gz_compact_rows16", "gz_expand_rows16", "gz_compact_block", "gz_expand_block",
    "gz_host_tables_create", "gz_host_tables_destroy", "gz_host_tables_array", "gz_host_tables_vocab_entry",
    "gz_host_tables_merge_entry", "gz_host_tables_symbol", "gz_limit", "gz_debug_set",
]

_lib = None


class GzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("genz_tokenize (HIP): error %d: %s" % (code, msg))
        self.code = code


def load_library():
    """dlopen the C-ABI library and declare its prototypes.  Raises RuntimeError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "genz_tokenize: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C genz-tokenize_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_size_t
    P = C.POINTER
    L.gz_version.restype = C.c_int
    L.gz_create.argtypes = [C.c_int, P(vp)]
    L.gz_destroy.argtypes = [vp]; L.gz_destroy.restype = None
    L.gz_last_error.argtypes = [vp]; L.gz_last_error.restype = C.c_char_p
    L.gz_load_tables.argtypes = [vp, vp, sz, vp, sz, P(C.c_char_p)]
    L.gz_table_info.argtypes = [vp, P(i32), P(i32), P(i32), P(i32)]
    L.gz_table_cache_status.argtypes = [vp]
    L.gz_table_digest.argtypes = [vp, vp]
    L.gz_vocab_entry.argtypes = [vp, i64, P(vp), P(i32), P(i32)]
    L.gz_merge_entry.argtypes = [vp, i64, P(vp), P(i32), P(i32), P(i32)]
    enc = [vp, vp, vp, vp, vp, i64, i32, u32, i64, vp, vp, vp, vp, vp, vp, vp, vp]
    L.gz_encode_batch.argtypes = enc
    L.gz_encode_batch_device.argtypes = enc
    L.gz_encode_batch_device_h.argtypes = enc + [vp, vp]
    L.gz_encode_batch_csr.argtypes = [vp, vp, vp, i64, i32, u32, vp, i64, i32, vp, P(i64)]
    L.gz_host_alloc.argtypes = [vp, sz, P(vp)]
    L.gz_host_free.argtypes = [vp, vp]
    L.gz_sync.argtypes = [vp]
    L.gz_word_token_counts.argtypes = [vp, C.c_int, vp, i64, vp, P(i64)]
    L.gz_bpe_word.argtypes = [vp, vp, i64, vp, i64]; L.gz_bpe_word.restype = i64
    L.gz_symbol_utf8.argtypes = [vp, i32, P(vp), P(i32)]
    L.gz_device_alloc.argtypes = [vp, sz, P(vp)]
    L.gz_device_free.argtypes = [vp, vp]
    L.gz_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    L.gz_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    L.gz_timing.argtypes = [vp, P(C.c_double)]
    L.gz_timing_history.argtypes = [vp, P(C.c_double), i32, P(i32)]
    L.gz_decoder_snapshot.argtypes = [vp]
    L.gz_decode_batch.argtypes = [vp, vp, vp, i64, vp, i32, vp, i64, vp]
    L.gz_decode_batch_device.argtypes = [vp, vp, vp, i64, vp, i32, vp, i64, vp, P(i64)]
    L.gz_preprocess_batch.argtypes = [vp, vp, i32, vp, vp, i64, vp, i64, vp]
    L.gz_preprocess_batch_device.argtypes = [vp, vp, i32, vp, vp, i64, i64, vp, i64, vp, P(i64)]
    L.gz_block_create.argtypes = [vp, vp, P(vp)]
    L.gz_block_release.argtypes = [vp]; L.gz_block_release.restype = None
    L.gz_block_dlpack.argtypes = [vp, i32, vp, i32, i32]; L.gz_block_dlpack.restype = vp
    L.gz_dlpack_capsule_destructor.argtypes = [vp]; L.gz_dlpack_capsule_destructor.restype = None
    L.gz_exchange_select.argtypes = [vp, C.c_int]
    L.gz_encode_emit_block.argtypes = [vp, vp, i32]
    L.gz_block_total.argtypes = [vp, i32, P(i64)]
    L.gz_comm_unique_id.argtypes = [vp]
    L.gz_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.gz_gather_rows.argtypes = [vp, vp, i64, i32, vp, vp, C.c_int]
    if hasattr(L, "gz_exchange_timing_history"):
        L.gz_exchange_timing_history.argtypes = [vp, P(C.c_double), i32, P(i32)]
    L.gz_compact_rows.argtypes = [vp, vp, vp, i64, i32, vp, P(i64)]
    L.gz_expand_rows.argtypes = [vp, vp, vp, i64, i32, vp, vp]
    L.gz_compact_rows16.argtypes = [vp, vp, vp, i64, i32, vp, P(i64)]
    L.gz_expand_rows16.argtypes = [vp, vp, vp, i64, i32, vp, vp]
    L.gz_compact_block.argtypes = [vp, vp, vp, i64, i32, i32, vp, P(i64)]
    L.gz_expand_block.argtypes = [vp, vp, i32, i64, i32, i64, vp, vp]
    L.gz_host_tables_create.argtypes = [vp, sz, vp, sz, P(C.c_char_p), P(vp)]
    L.gz_host_tables_destroy.argtypes = [vp]; L.gz_host_tables_destroy.restype = None
    L.gz_host_tables_array.argtypes = [vp, C.c_int, P(vp), P(i64)]
    L.gz_host_tables_vocab_entry.argtypes = [vp, i64, P(vp), P(i32), P(i32)]
    L.gz_host_tables_merge_entry.argtypes = [vp, i64, P(vp), P(i32), P(i32), P(i32)]
    L.gz_host_tables_symbol.argtypes = [vp, i32, P(vp), P(i32)]
    for name in SYMBOLS:
        if os.environ.get("GZ_LIBRARY") and not hasattr(L, name):
            continue                                     # (an older build loaded for an A/B run: entry points it lacks stay unbound)
        fn = getattr(L, name)
        if name not in ("gz_destroy", "gz_last_error", "gz_bpe_word", "gz_host_tables_destroy", "gz_block_release", "gz_block_dlpack",
                        "gz_dlpack_capsule_destructor", "gz_limit"):
            fn.restype = C.c_int
    if hasattr(L, "gz_limit"):
        L.gz_limit.argtypes = [C.c_int]
        L.gz_limit.restype = i64
    if hasattr(L, "gz_debug_set"):
        L.gz_debug_set.argtypes = [vp, C.c_char_p, i64]
    _lib = L
    return L


def debug_set(key: str, value: int, ctx: "Context | None" = None):
    """gz_debug_set: a test / experiment switch of the library -- of one context, or (ctx None) the process-wide defaults that
    contexts created afterwards copy and the table builder reads.  Raises ValueError for an unknown key or a bad value."""
    lib = load_library()
    rc = lib.gz_debug_set(ctx.handle if ctx is not None else None, key.encode("ascii"), int(value))
    if rc != GZ_OK:
        raise ValueError("gz_debug_set(%r, %r): unknown key or value out of range" % (key, value))


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def default_device() -> int:
    for k in ("GENZ_TOKENIZE_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(k)
        if v is not None and v.strip() != "":
            return int(v)
    return 0


class Context:
    """One gz_ctx: one GPU, one set of tables."""

    def __init__(self, device: int | None = None):
        self.lib = load_library()
        self.handle = C.c_void_p()
        self.device = default_device() if device is None else int(device)
        rc = self.lib.gz_create(self.device, C.byref(self.handle))
        if rc != GZ_OK:
            msg = self.lib.gz_last_error(None).decode("utf-8", "replace")
            self.handle = C.c_void_p()
            raise RuntimeError("genz_tokenize: cannot create a HIP context on device %d (%d: %s). "
                               "This build has no CPU fallback." % (self.device, rc, msg))

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.gz_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _check(self, rc):
        if rc < 0:
            raise GzError(rc, self.lib.gz_last_error(self.handle).decode("utf-8", "replace"))
        return rc

    # ---- tables ------------------------------------------------------------------------------------------
    def load_tables(self, vocab: bytes, bpe: bytes, specials):
        arr = (C.c_char_p * 5)(*[s.encode("utf-8", "surrogatepass") for s in specials])
        vb = C.create_string_buffer(vocab, len(vocab)) if vocab else None
        bb = C.create_string_buffer(bpe, len(bpe)) if bpe else None
        self._check(self.lib.gz_load_tables(self.handle, C.cast(vb, C.c_void_p) if vb else None, len(vocab),
                                            C.cast(bb, C.c_void_p) if bb else None, len(bpe), arr))

    def table_cache_status(self) -> int:
        """What the last load_tables did with the table cache: 0 no cache, 1 hit, 2 miss (written), 3 a file was refused."""
        return int(self.lib.gz_table_cache_status(self.handle))

    def table_digest(self) -> bytes:
        """SHA-256 over the device-resident table images and the dictionaries (gz_table_digest)."""
        out = (C.c_uint8 * 32)()
        self._check(self.lib.gz_table_digest(self.handle, out))
        return bytes(out)

    def table_info(self):
        vs, nr, ns = C.c_int32(), C.c_int32(), C.c_int32()
        sp = (C.c_int32 * 5)()
        self._check(self.lib.gz_table_info(self.handle, C.byref(vs), sp, C.byref(nr), C.byref(ns)))
        return vs.value, list(sp), nr.value, ns.value

    def vocab_items(self):
        n = self.table_info()[0]
        p, ln, idv = C.c_void_p(), C.c_int32(), C.c_int32()
        out = []
        for i in range(n):
            self._check(self.lib.gz_vocab_entry(self.handle, i, C.byref(p), C.byref(ln), C.byref(idv)))
            out.append((C.string_at(p, ln.value).decode("utf-8"), idv.value))
        return out

    def merge_items(self):
        n = self.table_info()[2]
        p, ln, nf, rk = C.c_void_p(), C.c_int32(), C.c_int32(), C.c_int32()
        out = []
        for i in range(n):
            self._check(self.lib.gz_merge_entry(self.handle, i, C.byref(p), C.byref(ln), C.byref(nf), C.byref(rk)))
            s = C.string_at(p, ln.value).decode("utf-8")
            out.append((tuple(s.split("\n")) if nf.value else (), rk.value))
        return out

    def symbol(self, sym: int) -> str:
        p, ln = C.c_void_p(), C.c_int32()
        self._check(self.lib.gz_symbol_utf8(self.handle, sym, C.byref(p), C.byref(ln)))
        return C.string_at(p, ln.value).decode("utf-8")

    # ---- encode, host buffers ------------------------------------------------------------------------------
    def encode(self, text: np.ndarray, text_off: np.ndarray, pair, pair_off, max_len, padding, truncation,
               extra_flags: int = 0):
        """Returns dict(input_ids, attention_mask, [token_type_ids, sequence_id, pair_len, status], row_off,
        n_real, dense).  Flat int32 arrays + int64 row offsets."""
        n = len(text_off) - 1
        flags = (GZ_PADDING if padding else 0) | (GZ_TRUNCATION if truncation else 0) | extra_flags
        ml = 0
        if max_len is None:
            flags |= GZ_MAX_LEN_NONE
        else:
            ml = int(max_len)
            if not -2**31 <= ml < 2**31:
                raise OverflowError("max_len does not fit in int32")
        is_pair = pair_off is not None
        dense = bool(max_len is not None and padding and truncation and ml >= 1)
        tb = int(text_off[-1] - text_off[0]) if n else 0
        pb = int(pair_off[-1] - pair_off[0]) if (is_pair and n) else 0
        if dense:
            cap = n * ml
        else:
            cap = tb + pb + (4 if is_pair else 2) * n
            if max_len is not None and padding and ml > 0:
                cap += n * ml
        ids = np.empty(max(cap, 1), dtype=np.int32)
        mask = np.empty(max(cap, 1), dtype=np.int32)
        tt = np.empty(max(cap, 1), dtype=np.int32) if is_pair else None
        seq = np.empty(max(cap, 1), dtype=np.int32) if is_pair else None
        row_off = np.zeros(n + 1, dtype=np.int64)
        pair_len = np.zeros(2 * max(n, 1), dtype=np.int32) if is_pair else None
        n_real = np.zeros(max(n, 1), dtype=np.int32)
        status = np.zeros(max(n, 1), dtype=np.int32)
        text = np.ascontiguousarray(text, dtype=np.uint8)
        text_off = np.ascontiguousarray(text_off, dtype=np.int64)
        if is_pair:
            pair = np.ascontiguousarray(pair, dtype=np.uint8)
            pair_off = np.ascontiguousarray(pair_off, dtype=np.int64)
        self._check(self.lib.gz_encode_batch(
            self.handle, _ptr(text), _ptr(text_off), _ptr(pair) if is_pair else None,
            _ptr(pair_off) if is_pair else None, n, ml, flags, cap, _ptr(ids), _ptr(mask), _ptr(tt), _ptr(seq),
            _ptr(row_off), _ptr(pair_len), _ptr(n_real), _ptr(status)))
        total = int(row_off[-1]) if n else 0
        out = dict(input_ids=ids[:total], attention_mask=mask[:total], row_off=row_off, n_real=n_real[:n],
                   status=status[:n], dense=dense, max_len=ml)
        if is_pair:
            out.update(token_type_ids=tt[:total], sequence_id=seq[:total], pair_len=pair_len[:2 * n].reshape(n, 2))
        return out

    # ---- host buffers in, CSR out, copies overlapped with the kernels ---------------------------------------------------
    def pinned_empty(self, shape, dtype=np.uint8) -> np.ndarray:
        """A numpy array in page-locked host memory (gz_host_alloc): H2D / D2H copies of it are real DMA.  The memory is
        returned when the array (and every view of it) is gone."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape, dtype=np.int64))
        p = C.c_void_p()
        self._check(self.lib.gz_host_alloc(self.handle, max(n * dtype.itemsize, 1), C.byref(p)))
        buf = (C.c_char * max(n * dtype.itemsize, 1)).from_address(p.value)
        lib, addr = self.lib, p.value
        import weakref
        arr = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)
        # (gz_host_free does not look at the context: the array may well outlive it -- close(), interpreter exit)
        weakref.finalize(buf, lambda: lib.gz_host_free(None, C.c_void_p(addr)))
        return arr

    def encode_csr(self, text: np.ndarray, text_off: np.ndarray, max_len: int, bits: int = 16, extra_flags: int = 0,
                   tokens: np.ndarray | None = None, n_real: np.ndarray | None = None):
        """gz_encode_batch_csr: (tokens[total] uint16|int32, n_real[N] int32).  `tokens` / `n_real` may be supplied
        (e.g. pinned); tokens must hold min(N * max_len, bytes + 2 N) entries to be safe."""
        n = len(text_off) - 1
        text = np.ascontiguousarray(text, dtype=np.uint8)
        text_off = np.ascontiguousarray(text_off, dtype=np.int64)
        tb = int(text_off[-1] - text_off[0]) if n else 0
        cap = min(n * int(max_len), tb + 2 * n)
        dt = np.uint16 if bits == 16 else np.int32
        if tokens is None:
            tokens = np.empty(max(cap, 1), dtype=dt)
        if n_real is None:
            n_real = np.empty(max(n, 1), dtype=np.int32)
        assert tokens.dtype == dt and n_real.dtype == np.int32 and tokens.flags["C_CONTIGUOUS"] and n_real.flags["C_CONTIGUOUS"]
        total = C.c_int64()
        self._check(self.lib.gz_encode_batch_csr(self.handle, _ptr(text), _ptr(text_off), n, int(max_len),
                                                 GZ_PADDING | GZ_TRUNCATION | extra_flags, _ptr(tokens), tokens.size, bits,
                                                 _ptr(n_real), C.byref(total)))
        return tokens[:total.value], n_real[:n]

    def word_token_counts(self, which_text: int, n_docs: int, capacity: int):
        """(counts[int32, words], doc_first[int64, n_docs+1]) of the last encode call."""
        counts = np.empty(max(capacity, 1), dtype=np.int32)
        first = np.zeros(n_docs + 1, dtype=np.int64)
        nw = C.c_int64()
        rc = self.lib.gz_word_token_counts(self.handle, which_text, _ptr(counts), capacity, _ptr(first), C.byref(nw))
        if rc == GZ_E_CAPACITY and nw.value > capacity:         # the call says how many words there are: once more, exactly
            capacity = int(nw.value)
            counts = np.empty(capacity, dtype=np.int32)
            rc = self.lib.gz_word_token_counts(self.handle, which_text, _ptr(counts), capacity, _ptr(first), C.byref(nw))
        self._check(rc)
        return counts[:nw.value], first

    def bpe_word(self, word: bytes):
        cap = len(word) + 1
        out = np.empty(cap, dtype=np.int32)
        buf = np.frombuffer(word, dtype=np.uint8)
        n = self.lib.gz_bpe_word(self.handle, _ptr(buf), len(word), _ptr(out), cap)
        self._check(n)
        return out[:n]

    # ---- device-resident path (bench, multi-GPU) ---------------------------------------------------------------
    def alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self.lib.gz_device_alloc(self.handle, nbytes, C.byref(p)))
        return p.value

    def free(self, dptr: int):
        self._check(self.lib.gz_device_free(self.handle, C.c_void_p(dptr)))

    def h2d(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._check(self.lib.gz_memcpy_h2d(self.handle, C.c_void_p(dptr), _ptr(arr), arr.nbytes))

    def d2h(self, arr: np.ndarray, dptr: int):
        assert arr.flags["C_CONTIGUOUS"]
        self._check(self.lib.gz_memcpy_d2h(self.handle, _ptr(arr), C.c_void_p(dptr), arr.nbytes))

    def encode_device(self, d_text, d_text_off, d_pair, d_pair_off, n_docs, max_len, flags, capacity,
                      d_ids, d_mask, d_tt=None, d_seq=None, d_row_off=None, d_pair_len=None, d_n_real=None,
                      d_status=None, h_text_off=None, h_pair_off=None):
        """Enqueue one call on device-resident buffers.  h_text_off / h_pair_off: optional host copies (int64 numpy) of
        the offsets; with them the library does not read the batch's byte sizes back from the device first."""
        vp = C.c_void_p
        args = [self.handle, vp(d_text), vp(d_text_off), vp(d_pair) if d_pair else None,
                vp(d_pair_off) if d_pair_off else None, n_docs, max_len, flags, capacity, vp(d_ids), vp(d_mask),
                vp(d_tt) if d_tt else None, vp(d_seq) if d_seq else None, vp(d_row_off) if d_row_off else None,
                vp(d_pair_len) if d_pair_len else None, vp(d_n_real) if d_n_real else None,
                vp(d_status) if d_status else None]
        if h_text_off is not None:
            self._check(self.lib.gz_encode_batch_device_h(*args, _ptr(h_text_off), _ptr(h_pair_off)))
        else:
            self._check(self.lib.gz_encode_batch_device(*args))

    def sync(self):
        self._check(self.lib.gz_sync(self.handle))

    def timing_history(self, max_calls: int = 64):
        """Synchronise; whole-pipeline kernel time (ms) of the last timed calls, oldest first."""
        buf = (C.c_double * max_calls)()
        n = C.c_int32()
        self._check(self.lib.gz_timing_history(self.handle, buf, max_calls, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    def exchange_timing_history(self, max_calls: int = 64):
        """Synchronise the exchange stream; start-possible -> done time (ms) of the last gz_gather_rows calls, oldest first."""
        buf = (C.c_double * max_calls)()
        n = C.c_int32()
        self._check(self.lib.gz_exchange_timing_history(self.handle, buf, max_calls, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    def timing(self):
        t = (C.c_double * 4)()
        self._check(self.lib.gz_timing(self.handle, t))
        return list(t)

    # ---- RCCL ---------------------------------------------------------------------------------------------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self._check(self.lib.gz_comm_unique_id(C.cast(buf, C.c_void_p)))
        return buf.raw

    def comm_init(self, uid: bytes, rank: int, world: int):
        buf = C.create_string_buffer(uid, 128)
        self._check(self.lib.gz_comm_init(self.handle, C.cast(buf, C.c_void_p), rank, world))

    # ---- batch decode -----------------------------------------------------------------------------------
    def decoder_snapshot(self):
        self._check(self.lib.gz_decoder_snapshot(self.handle))

    def decode(self, ids: np.ndarray, row_off: np.ndarray, unk: bytes):
        """ids int32 (packed rows), row_off int64 [n+1]  ->  (bytes of all rows, out_off int64 [n+1])."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        row_off = np.ascontiguousarray(row_off, dtype=np.int64)
        n = len(row_off) - 1
        out_off = np.zeros(n + 1, dtype=np.int64)
        ub = C.create_string_buffer(unk, len(unk))
        cap = max(64, 6 * len(ids))
        for _ in range(2):
            out = np.empty(cap, dtype=np.uint8)
            rc = self.lib.gz_decode_batch(self.handle, _ptr(ids) if len(ids) else None, _ptr(row_off), n,
                                          C.cast(ub, C.c_void_p), len(unk), _ptr(out), cap, _ptr(out_off))
            if rc != GZ_E_CAPACITY:
                break
            cap = int(out_off[n])
        self._check(rc)
        return out[:int(out_off[n])], out_off

    def decode_device(self, d_ids, d_row_off, n_rows, unk: bytes, d_out, capacity, d_out_off) -> int:
        total = C.c_int64()
        ub = C.create_string_buffer(unk, len(unk))
        self._check(self.lib.gz_decode_batch_device(self.handle, C.c_void_p(d_ids), C.c_void_p(d_row_off), n_rows,
                                                    C.cast(ub, C.c_void_p), len(unk), C.c_void_p(d_out) if d_out else None,
                                                    capacity, C.c_void_p(d_out_off), C.byref(total)))
        return total.value

    # ---- text pre-pass ----------------------------------------------------------------------------------
    def preprocess(self, ops, text: np.ndarray, text_off: np.ndarray):
        """packed UTF-8 + int64 offsets -> (packed UTF-8, int64 offsets) after applying the GZ_PP_* filters in order."""
        ops = np.ascontiguousarray(ops, dtype=np.int32)
        text = np.ascontiguousarray(text, dtype=np.uint8)
        text_off = np.ascontiguousarray(text_off, dtype=np.int64)
        n = len(text_off) - 1
        cap = int(text_off[n] - text_off[0])
        out = np.empty(max(cap, 1), dtype=np.uint8)
        out_off = np.zeros(n + 1, dtype=np.int64)
        self._check(self.lib.gz_preprocess_batch(self.handle, _ptr(ops), len(ops), _ptr(text) if len(text) else None,
                                                 _ptr(text_off), n, _ptr(out), cap, _ptr(out_off)))
        return out[:int(out_off[n])], out_off

    def preprocess_device(self, ops, d_text, d_off, n_docs, text_bytes, d_out, capacity, d_out_off) -> int:
        ops = np.ascontiguousarray(ops, dtype=np.int32)
        total = C.c_int64()
        self._check(self.lib.gz_preprocess_batch_device(self.handle, _ptr(ops), len(ops), C.c_void_p(d_text), C.c_void_p(d_off),
                                                        n_docs, text_bytes, C.c_void_p(d_out) if d_out else None, capacity,
                                                        C.c_void_p(d_out_off), C.byref(total)))
        return total.value

    def exchange_select(self, back: int):
        """Exchange operations issued from now on belong to the encode call `back` calls before the latest one."""
        self._check(self.lib.gz_exchange_select(self.handle, back))

    def compact_rows(self, d_rows, d_n_real, n_rows, row_len, d_out, bits: int = 32) -> int:
        """Rows without their padding, back to back, as int32 (bits=32) or uint16 (bits=16) entries; returns the count."""
        total = C.c_int64()
        fn = self.lib.gz_compact_rows16 if bits == 16 else self.lib.gz_compact_rows
        self._check(fn(self.handle, C.c_void_p(d_rows), C.c_void_p(d_n_real), n_rows, row_len, C.c_void_p(d_out), C.byref(total)))
        return total.value

    def expand_rows(self, d_compact, d_n_real, n_rows, row_len, d_ids, d_mask, bits: int = 32):
        fn = self.lib.gz_expand_rows16 if bits == 16 else self.lib.gz_expand_rows
        self._check(fn(self.handle, C.c_void_p(d_compact), C.c_void_p(d_n_real), n_rows, row_len, C.c_void_p(d_ids), C.c_void_p(d_mask)))

    def encode_emit_block(self, d_block, bits: int = 16):
        """Arms the next (dense) encode call to leave its exchange block in d_block as part of the call (0: disarm)."""
        self._check(self.lib.gz_encode_emit_block(self.handle, C.c_void_p(d_block) if d_block else None, bits))

    def block_total(self, back: int = 0) -> int:
        """Waits for the encode call `back` calls ago; the number of entries of the block it emitted."""
        total = C.c_int64()
        self._check(self.lib.gz_block_total(self.handle, back, C.byref(total)))
        return total.value

    def compact_block(self, d_rows, d_n_real, n_rows, row_len, d_block, bits: int = 16) -> int:
        """One rank's message of the exchange step: [int32 n_real[n_rows] | uint32 first[n_rows] | the rows' real entries, `bits`
        bits each] into d_block; returns the number of entries."""
        total = C.c_int64()
        self._check(self.lib.gz_compact_block(self.handle, C.c_void_p(d_rows), C.c_void_p(d_n_real), n_rows, row_len, bits,
                                              C.c_void_p(d_block), C.byref(total)))
        return total.value

    def expand_block(self, d_block, n_rows, row_len, d_ids, d_mask, bits: int = 16, total: int = 0xFFFFFFFF):
        """`total`: the entry count the block was announced with (block_total / compact_block on the sending side)."""
        self._check(self.lib.gz_expand_block(self.handle, C.c_void_p(d_block), bits, n_rows, row_len, int(total), C.c_void_p(d_ids), C.c_void_p(d_mask)))

    def gather_rows(self, d_send, n_rows_local, row_len, d_recv, rows_per_rank, root=0):
        rpr = np.ascontiguousarray(rows_per_rank, dtype=np.int64)
        self._check(self.lib.gz_gather_rows(self.handle, C.c_void_p(d_send), n_rows_local, row_len,
                                            C.c_void_p(d_recv) if d_recv else None, _ptr(rpr), root))


class HostTables:
    """gz_host_tables: the loader + table builder run on the host only (no GPU).  Used by the CPU test-suite to
    check the integer tables the kernels consume, and by tools that want to inspect them."""

    _DT = {0: (np.uint32, 4), 1: (np.uint32, 4), 2: (np.int32, 2), 3: (np.uint32, 2), 4: (np.uint32, 4), 5: (np.int32, 1),
           6: (np.uint32, 2), 7: (np.uint16, 1), 8: (np.uint32, 1), 9: (np.uint32, 2)}

    def __init__(self, vocab: bytes, bpe: bytes, specials=("<pad>", "<s>", "</s>", "<mask>", "<unk>")):
        self.lib = load_library()
        self.handle = C.c_void_p()
        arr = (C.c_char_p * 5)(*[s.encode("utf-8", "surrogatepass") for s in specials])
        vb = C.create_string_buffer(vocab, len(vocab)) if vocab else None
        bb = C.create_string_buffer(bpe, len(bpe)) if bpe else None
        rc = self.lib.gz_host_tables_create(C.cast(vb, C.c_void_p) if vb else None, len(vocab),
                                            C.cast(bb, C.c_void_p) if bb else None, len(bpe), arr, C.byref(self.handle))
        if rc != GZ_OK:
            raise GzError(rc, self.lib.gz_last_error(None).decode("utf-8", "replace"))

    def array(self, which: int) -> np.ndarray:
        p, n = C.c_void_p(), C.c_int64()
        rc = self.lib.gz_host_tables_array(self.handle, which, C.byref(p), C.byref(n))
        if rc != GZ_OK:
            raise GzError(rc, "gz_host_tables_array")
        dt, w = self._DT[which]
        if n.value == 0:
            return np.zeros((0, w) if w > 1 else 0, dtype=dt)
        buf = (C.c_char * (n.value * w * np.dtype(dt).itemsize)).from_address(p.value)
        a = np.frombuffer(buf, dtype=dt).copy()
        return a.reshape(n.value, w) if w > 1 else a

    def vocab_items(self):
        out, i = [], 0
        p, ln, idv = C.c_void_p(), C.c_int32(), C.c_int32()
        while self.lib.gz_host_tables_vocab_entry(self.handle, i, C.byref(p), C.byref(ln), C.byref(idv)) == GZ_OK:
            out.append((C.string_at(p, ln.value).decode("utf-8"), idv.value)); i += 1
        return out

    def merge_items(self):
        out, i = [], 0
        p, ln, nf, rk = C.c_void_p(), C.c_int32(), C.c_int32(), C.c_int32()
        while self.lib.gz_host_tables_merge_entry(self.handle, i, C.byref(p), C.byref(ln), C.byref(nf), C.byref(rk)) == GZ_OK:
            s = C.string_at(p, ln.value).decode("utf-8")
            out.append((tuple(s.split("\n")) if nf.value else (), rk.value)); i += 1
        return out

    def symbols(self):
        out, i = [], 0
        p, ln = C.c_void_p(), C.c_int32()
        while self.lib.gz_host_tables_symbol(self.handle, i, C.byref(p), C.byref(ln)) == GZ_OK:
            out.append(C.string_at(p, ln.value).decode("utf-8")); i += 1
        return out

    def close(self):
        if self.handle.value:
            self.lib.gz_host_tables_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
