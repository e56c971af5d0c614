"""MI355X-native drop-in for the tokenizer of DVNghiem/genz-tokenize (reference genz_tokenize/__init__.py:1-10
exports `Tokenize`).  `genz_tokenize.preprocess` mirrors the reference's text filters on the GPU; the reference's
`models` and `ranking` sub-packages are out of scope, see DESIGN.md."""
from .tokenize import Tokenize, get_pairs  # noqa: F401

__all__ = ['Tokenize', 'get_pairs']
__version__ = '1.2.7+mi355x.1'
