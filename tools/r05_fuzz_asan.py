import sys, os, random, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gz_oracle_c as OC
OC._LIB = os.environ["GZ_ORACLE_ASAN"]     # gcc -O1 -g -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared oracle/gz_oracle.c
import host_asan_child as H   # loads the asan host lib

def make(seed):
    r = random.Random(1000 + seed)
    alphabet = list("ab") + r.sample(list("cdeghk_"), 3) + r.sample(["â", "ệ", "đ", "\U0001F600", "中"], 2)
    if seed % 2 == 0:
        alphabet += ["<", "/", "w", ">", "@"]
    syms = list(alphabet); finals = [a + "</w>" for a in alphabet]; merges = []
    for _ in range(r.choice([40, 150, 600, 1500])):
        left = r.choice(syms)
        if r.random() < 0.35:
            right = r.choice(finals); new = left + right; finals.append(new)
        else:
            right = r.choice(syms); new = left + right; syms.append(new)
        if len(new) > 40: continue
        merges.append(left + " " + right)
    if r.random() < 0.5: merges.insert(0, "#version: 0.2")
    if seed == 3: merges += merges[5:25]
    bpe = ("\n".join(merges) + "\n").encode("utf-8")
    words = set()
    for x in syms:
        if r.random() < 0.8: words.add(x + "@@")
    for x in finals:
        if r.random() < 0.8: words.add(x[:-4])
    words = sorted(words); r.shuffle(words)
    vocab = "".join("%s %d\n" % (w, r.randint(1, 99)) for w in words).encode("utf-8")
    def word():
        n = r.choice([1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 17, 20, 31, 32, 33, 50, 64, 65, 100, 300])
        return "".join(r.choice(alphabet) for _ in range(n))
    docs = []
    for _ in range(1500):
        k = r.choice([0, 1, 2, 5, 10, 30])
        docs.append("".join(word() + r.choice([" ", " ", "  ", "\n", "\n ", "\t", "　"]) for _ in range(k)))
    return vocab, bpe, docs, len(merges)

for seed in range(1, 13):
    vocab, bpe, docs, nm = make(seed)
    rc, voc, mer = H.build(vocab, bpe)
    co = OC.COracle(vocab, bpe)
    text, toff = OC._pack(docs)
    text = np.ascontiguousarray(text)
    tot = 0
    for ml, pad, tr in ((None, True, True), (24, True, True)):
        ids, mask, _, _, row, _, _ = co.call_packed(text, toff, max_len=ml, padding=pad, truncation=tr)
        tot += int(row[-1])
    del co
    print("seed", seed, "merges", nm, "vocab", len(voc), "host rc", rc, "text bytes", int(toff[-1]), "docs", len(docs), "tokens", tot, flush=True)
print("ASAN FUZZ OK")
