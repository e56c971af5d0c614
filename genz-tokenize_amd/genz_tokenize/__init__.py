"""MI355X-native drop-in for the tokenizer of DVNghiem/genz-tokenize (reference genz_tokenize/__init__.py:1-10
exports `Tokenize`; the reference's `preprocess` and `models` sub-packages are out of scope)."""
from .tokenize import Tokenize, get_pairs  # noqa: F401

__all__ = ['Tokenize', 'get_pairs']
__version__ = '1.2.7+mi355x.1'
