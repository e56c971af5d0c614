"""Child of tests/test_host_tables.py::test_host_builder_under_sanitizers: loads the ASan + UBSan build of the host
table builder (make -C genz-tokenize_amd/csrc asan) with plain ctypes and runs, through the C ABI's gz_host_tables_*:
  1. the bundled tables (48 k vocab lines, 50 k merges),
  2. every G4 loader fixture (BOM, CR / CRLF, missing space, duplicates, 3-field merges, invalid UTF-8, ...) with the
     results compared against the fixture's `encoder` / `bpe_ranks`,
  3. a byte-level fuzz: the G4 tables and slices of the bundled files with random bytes flipped, inserted, deleted,
     truncated, and with random special-token strings.
Any out-of-bounds access, use-after-free or undefined behaviour aborts the process (the parent checks the exit code
and stderr).  Run with LD_PRELOAD=libasan (the parent does)."""
import base64
import ctypes as C
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "genz-tokenize_amd", "genz_tokenize", "libgenz_tokenize_host_asan.so")
DATA = os.path.join(ROOT, "genz-tokenize_amd", "genz_tokenize", "data")
L = C.CDLL(LIB)
vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
L.gz_host_tables_create.argtypes = [vp, sz, vp, sz, C.POINTER(C.c_char_p), C.POINTER(vp)]
L.gz_host_tables_destroy.argtypes = [vp]; L.gz_host_tables_destroy.restype = None
L.gz_host_tables_array.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(i64)]
L.gz_host_tables_vocab_entry.argtypes = [vp, i64, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32)]
L.gz_host_tables_merge_entry.argtypes = [vp, i64, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
L.gz_host_tables_symbol.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i32)]
L.gz_last_error.argtypes = [vp]; L.gz_last_error.restype = C.c_char_p
L.gz_debug_set.argtypes = [vp, C.c_char_p, i64]
sys.path.insert(0, ROOT)
import gz_switches                                              # GZ_TEST_SWITCHES -> gz_debug_set(NULL, ...) of THIS library
gz_switches.apply(lib=L)
SPECIALS = ("<pad>", "<s>", "</s>", "<mask>", "<unk>")
WIDTH = {0: 16, 1: 16, 2: 8, 3: 8, 4: 16, 5: 4, 6: 8, 7: 2, 8: 4, 9: 8}


def build(vocab: bytes, bpe: bytes, specials=SPECIALS, read_all=True):
    """Returns (rc, vocab items, merge items); walks every table the builder produced (so that ASan sees every byte)."""
    h = vp()
    arr = (C.c_char_p * 5)(*[s.encode("utf-8", "surrogatepass") for s in specials])
    vb = C.create_string_buffer(vocab, len(vocab)) if vocab else None
    bb = C.create_string_buffer(bpe, len(bpe)) if bpe else None
    rc = L.gz_host_tables_create(C.cast(vb, vp) if vb else None, len(vocab), C.cast(bb, vp) if bb else None, len(bpe), arr, C.byref(h))
    if rc != 0:
        assert L.gz_last_error(None) is not None
        return rc, None, None
    voc, mer = [], []
    p, ln, idv, nf, rk, n = vp(), i32(), i32(), i32(), i32(), i64()
    i = 0
    while L.gz_host_tables_vocab_entry(h, i, C.byref(p), C.byref(ln), C.byref(idv)) == 0:
        voc.append((C.string_at(p, ln.value), idv.value)); i += 1
    i = 0
    while L.gz_host_tables_merge_entry(h, i, C.byref(p), C.byref(ln), C.byref(nf), C.byref(rk)) == 0:
        mer.append((C.string_at(p, ln.value), nf.value, rk.value)); i += 1
    if read_all:
        i = 0
        while L.gz_host_tables_symbol(h, i, C.byref(p), C.byref(ln)) == 0:
            C.string_at(p, ln.value); i += 1
        for which in range(1, 10):                              # (0 was the linear-probing pair table of rounds 1-3)
            assert L.gz_host_tables_array(h, which, C.byref(p), C.byref(n)) == 0
            if n.value:
                C.string_at(p, n.value * WIDTH[which])               # touch every byte of the table
    L.gz_host_tables_destroy(h)
    return 0, voc, mer


def mutate(r, data: bytes) -> bytes:
    b = bytearray(data)
    for _ in range(r.choice([1, 1, 2, 3, 8])):
        k = r.random()
        pos = r.randrange(len(b) + 1)
        if k < 0.08 and b:
            b[min(pos, len(b) - 1)] = r.randrange(256)
        elif k < 0.3 and b:
            b[min(pos, len(b) - 1)] = r.choice(b" \n\rab_@<>/w.9")                       # stays valid UTF-8 most of the time
        elif k < 0.5:
            b[pos:pos] = bytes(r.choice([b"\n", b" ", b"\r", b"\r\n", b"\xef\xbb\xbf", b"\xc2\xa0", b"\xe3\x80\x80", b"\xf0\x9f\x98\x80",
                                        b"</w>", b"@@", b"\xff", b"\xc3", b"\xe1\xba", b"\x00", b"\t", b"#version: 0.2"]))
        elif k < 0.7 and b:
            del b[min(pos, len(b) - 1):min(pos, len(b) - 1) + r.choice([1, 1, 2, 7])]
        elif k < 0.8:
            b = b[:pos]
        elif k < 0.9:
            b = b + b[:r.randrange(len(b) + 1)]
        elif k < 0.93:
            b[pos:pos] = bytes(r.randrange(256) for _ in range(r.choice([1, 2, 3, 4, 17])))
        else:
            b[pos:pos] = "".join(r.choice("aăâbcdđeêghiklmnoôơ_ \n\u3000\U0001F600@</w>") for _ in range(r.choice([1, 3, 9, 40]))).encode()
    return bytes(b)


def main():
    r = random.Random(20261004)
    vocab = open(os.path.join(DATA, "vocab.txt"), "rb").read()
    bpe = open(os.path.join(DATA, "bpe.codes"), "rb").read()
    rc, voc, mer = build(vocab, bpe)
    assert rc == 0 and len(voc) == 48423 and len(mer) == 50001, (rc, len(voc or []), len(mer or []))
    n_g4 = n_fuzz = n_err = 0
    seeds = []
    for line in open(os.path.join(ROOT, "tests", "golden", "g4_loader.jsonl")):
        row = json.loads(line)
        v, b = base64.b64decode(row["vocab_b64"]), base64.b64decode(row["bpe_b64"])
        seeds.append((v, b))
        rc, voc, mer = build(v, b)
        if "calls" not in row:
            assert (rc == 0) == (row["raises"] is None), row["name"]
        else:
            assert rc == 0, row["name"]
            got = sorted(((w.decode("utf-8"), i) for w, i in voc), key=lambda kv: (kv[1], kv[0]))
            assert got == [tuple(x) for x in row["encoder"]], row["name"]                # (the fixture lists the dict sorted by id, word)
            assert len(mer) == len(row["bpe_ranks"]), row["name"]
        n_g4 += 1
    seeds.append((vocab[:4000], bpe[:4000]))
    seeds.append((vocab[300000:304000], bpe[200000:206000]))
    seeds.append((b"", b""))
    for it in range(int(os.environ.get("GZ_FUZZ_ITERS", "6000"))):
        v, b = seeds[it % len(seeds)]
        if r.random() < 0.8:
            v = mutate(r, v)
        if r.random() < 0.8:
            b = mutate(r, b)
        sp = SPECIALS
        if r.random() < 0.2:
            sp = tuple(r.choice(["<pad>", "", "a", "ấ", "\U0001F600", "</w>", "x@@", " ", "<unk>", "\ud800"]) for _ in range(5))
        rc, _, _ = build(v, b, sp, read_all=(it % 3 == 0))
        n_fuzz += 1
        n_err += rc != 0
    print("asan child ok: bundled tables, %d loader fixtures, %d fuzzed table pairs (%d refused as invalid)" % (n_g4, n_fuzz, n_err))


if __name__ == "__main__":
    main()
