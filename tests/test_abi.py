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
    assert lib.gz_version() == 0x010000


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
