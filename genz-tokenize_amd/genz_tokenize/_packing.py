"""list of str -> packed UTF-8 + offsets, the input form of every batch entry point of the C ABI.

`pack` is host-side plumbing, not tokenizer logic: csrc/gz_pack.c (`make pack`, a CPython extension) does it in C; when
that module is not built the Python loop below produces the same bytes, slower.  Encoding is UTF-8 with 'surrogatepass'
(a Python str may hold lone surrogates; the reference's `re` and dict lookups see them as ordinary characters)."""
from typing import Sequence

import numpy as np

try:
    from . import _gz_pack
except ImportError:                             # pragma: no cover
    _gz_pack = None


PINNED_MIN_TEXTS = 20000        # smaller batches are packed into an ordinary buffer: the library stages those through its own pinned blocks


def pack_pinned(texts: Sequence[str], holder, ctx):
    """`pack` straight into a page-locked arena `holder` keeps (holder._pin_text: grown when too small): no fresh buffer of the
    batch's size, no second copy, and the H2D copy that follows is real DMA.  Returns (uint8 view of the arena, int64 offsets).
    Falls back to `pack` when the C packer is not built, and for batches of fewer than PINNED_MIN_TEXTS texts (no 16 MB of pinned
    memory for a handful of sentences).

    The returned text is a VIEW of the one arena the holder keeps: the next call on the same holder overwrites it.  Callers hold
    `holder._batch_lock` from this call until the library has consumed the text (Tokenize.encode_batch / encode_to_device do)."""
    if _gz_pack is None or not hasattr(_gz_pack, "pack_into") or len(texts) < PINNED_MIN_TEXTS:
        return pack(texts)
    off = np.empty(len(texts) + 1, dtype=np.int64)
    arena = getattr(holder, "_pin_text", None)
    if arena is None:
        arena = holder._pin_text = ctx.pinned_empty(1 << 24, np.uint8)
    n = _gz_pack.pack_into(texts, arena, off)
    if n < 0:                                        # too small: the call said how much it takes
        arena = holder._pin_text = None
        arena = holder._pin_text = ctx.pinned_empty(int(-n * 1.25) + 4096, np.uint8)
        n = _gz_pack.pack_into(texts, arena, off)
    return arena[:n], off


def pack(texts: Sequence[str]):
    """(uint8 buffer, int64 offsets[N + 1]); TypeError("expected string or bytes-like object") for a non-str item --
    what `re.findall` raises in the reference (tokenize.py:106)."""
    if _gz_pack is not None and len(texts) >= 8:
        b, o = _gz_pack.pack(texts)
        return np.frombuffer(b, dtype=np.uint8), np.frombuffer(o, dtype=np.int64)
    parts = []
    for t in texts:
        if not isinstance(t, str):
            raise TypeError("expected string or bytes-like object")
        parts.append(t.encode("utf-8", "surrogatepass"))
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    if parts:
        np.cumsum([len(p) for p in parts], out=off[1:])
    buf = np.frombuffer(b"".join(parts), dtype=np.uint8) if off[-1] else np.zeros(0, dtype=np.uint8)
    return buf, off
