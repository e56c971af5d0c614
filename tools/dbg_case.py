import gzip, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
from genz_tokenize import Tokenize
sys.path.insert(0, ROOT); import gz_switches; gz_switches.apply()      # GZ_TEST_SWITCHES="key=value,..." -> gz_debug_set (the library reads no switch from the environment)
tok = Tokenize()
rows = [json.loads(l) for l in gzip.open(os.path.join(ROOT, "tests/golden/g3_random.jsonl.gz"), "rt")]
bad = 0
for r in rows:
    if "raises" in r or not r["kwargs"].get("return_offset"):
        continue
    args = list(r["args"])
    got = tok(*args, **r["kwargs"])
    got = json.loads(json.dumps(got))
    if got != r["result"]:
        bad += 1
        if bad <= 3:
            go, wo = got["offset"], r["result"]["offset"]
            k = next(i for i in range(min(len(go), len(wo))) if go[i] != wo[i])
            words = args[0].split()
            print("first diff at offset entry", k, "got", go[k:k+3], "want", wo[k:k+3], "n entries", len(go), len(wo))
            print("word:", repr(words[k-1]) if 0 < k <= len(words) else None, "bytes", len(words[k-1].encode()) if 0 < k <= len(words) else None, "chars", len(words[k-1]) if 0 < k <= len(words) else None)
print("bad", bad)
