"""Test / tool plumbing for the library's typed switches (gz_debug_set, include/genz_tokenize.h).

The library reads no switch from the environment.  Tests that re-run a selection of the suite in a child process, the A/B tools
and bench.py's diagnostic modes say what they want in ONE variable of their own,

    GZ_TEST_SWITCHES="small=0,scan_multi=0"

and call `apply()` once, before any context exists: every pair goes to gz_debug_set(NULL, key, value) -- the process-wide
defaults that contexts created afterwards copy and that the table builder reads.  Not part of the product package.
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def parse(text):
    out = []
    for item in (text or "").replace(";", ",").split(","):
        item = item.strip()
        if not item:
            continue
        k, _, v = item.partition("=")
        out.append((k.strip(), int(v.strip())))
    return out


def encode(**kv):
    """The value of GZ_TEST_SWITCHES for these switches."""
    return ",".join("%s=%d" % (k, int(v)) for k, v in kv.items())


def apply(text=None, lib=None):
    """Sets the switches named in `text` (default: $GZ_TEST_SWITCHES) as process-wide defaults.  Returns the pairs that were set."""
    pairs = parse(os.environ.get("GZ_TEST_SWITCHES") if text is None else text)
    if not pairs:
        return []
    if lib is None:
        pkg = os.path.join(ROOT, "genz-tokenize_amd")
        if pkg not in sys.path:
            sys.path.insert(0, pkg)
        from genz_tokenize import _native
        lib = _native.load_library()
    for k, v in pairs:
        rc = lib.gz_debug_set(None, k.encode("ascii"), v)
        if rc != 0:
            raise ValueError("GZ_TEST_SWITCHES: unknown key or value out of range: %s=%d" % (k, v))
    return pairs
