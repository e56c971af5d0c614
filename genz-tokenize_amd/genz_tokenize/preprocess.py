"""Drop-in for genz_tokenize/preprocess.py: the text filters users run before tokenizing, computed on the GPU.

Same function names and results as the reference (remove_html, convert_unicode, remove_punctuations, remove_emoji,
remove_URL); each call runs a batch of one document through the HIP pre-pass (csrc/gz_preproc.inc).  `preprocess_batch`
is the form meant for real work: many documents, several filters chained, one trip to the device.  There is no CPU
fallback.  `vncore_tokenize` (preprocess.py:83-89) only forwards to an external VnCoreNLP server and is not mirrored.
"""
from typing import List, Optional, Sequence

import numpy as np

from . import _native
from ._packing import pack as _pack

OPS = {"html": _native.GZ_PP_HTML, "unicode": _native.GZ_PP_UNICODE, "punct": _native.GZ_PP_PUNCT,
       "emoji": _native.GZ_PP_EMOJI, "url": _native.GZ_PP_URL}
_ctx: Optional[_native.Context] = None


def _context() -> _native.Context:
    global _ctx
    if _ctx is None:
        _ctx = _native.Context()
    return _ctx


def preprocess_packed(text: np.ndarray, offsets: np.ndarray, ops: Sequence[str], ctx: Optional[_native.Context] = None):
    """Packed UTF-8 bytes + int64 offsets in, the same out (ready for Tokenize.encode_packed)."""
    codes = [OPS[o] for o in ops]
    if not codes:
        return np.asarray(text, dtype=np.uint8), np.asarray(offsets, dtype=np.int64)
    return (ctx or _context()).preprocess(codes, text, offsets)


def preprocess_batch(texts: Sequence[str], ops: Sequence[str], ctx: Optional[_native.Context] = None) -> List[str]:
    """Apply the named filters ('html', 'unicode', 'punct', 'emoji', 'url'), in the given order, to every text."""
    buf, off = _pack(texts)
    out, oo = preprocess_packed(buf, off, ops, ctx)
    raw = out.tobytes()
    return [raw[oo[i]:oo[i + 1]].decode("utf-8", "surrogatepass") for i in range(len(texts))]


def remove_html(txt: str):
    '''Remove html tag (preprocess.py:5-9)'''
    return preprocess_batch([txt], ["html"])[0]


def convert_unicode(txt: str):
    '''Composed (base + combining tone mark) -> precomposed Vietnamese letters (preprocess.py:32-36)'''
    return preprocess_batch([txt], ["unicode"])[0]


def remove_punctuations(txt: str):
    '''Drop every string.punctuation character (preprocess.py:39-44)'''
    return preprocess_batch([txt], ["punct"])[0]


def remove_emoji(txt: str):
    '''Remove emoji, then collapse whitespace (preprocess.py:47-72)'''
    return preprocess_batch([txt], ["emoji"])[0]


def remove_URL(txt: str):
    '''remove url (preprocess.py:75-80)'''
    return preprocess_batch([txt], ["url"])[0]
