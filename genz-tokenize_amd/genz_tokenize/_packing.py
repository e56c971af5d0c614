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
