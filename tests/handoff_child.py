"""Child process of test_device_handoff_dlpack: torch first, then the tokenizer; DLPack hand-off checks."""
import gc
import os
import sys

import numpy as np
import torch

torch.cuda.init()
assert torch.cuda.is_available()
_ = torch.zeros(1, device="cuda")          # torch owns the HIP runtime of this process from here on

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "genz-tokenize_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import corpus  # noqa: E402
import gz_oracle as O  # noqa: E402
from genz_tokenize import Tokenize  # noqa: E402

tok = Tokenize()
t_or = O.Tables(open(corpus.VOCAB_PATH, "rb").read(), open(corpus.BPE_PATH, "rb").read())
text, offs, _ = corpus.config_corpus(2, n_docs=3000, seed=9)
raw = text.tobytes()
docs = [raw[offs[i]:offs[i + 1]].decode("utf-8") for i in range(len(offs) - 1)]
for pairs in (None, docs[::-1]):
    out = tok.encode_to_device(docs, pairs, max_len=48)
    host = tok.encode_batch(docs, pairs, max_len=48)
    for k in ("input_ids", "attention_mask") + (("token_type_ids", "sequence_id") if pairs else ()):
        t = torch.from_dlpack(out[k])
        assert t.is_cuda and t.dtype == torch.int32 and tuple(t.shape) == (len(docs), 48), k
        assert t.data_ptr() == out[k].ptr, "not zero-copy"
        if k in ("input_ids", "attention_mask"):
            assert np.array_equal(t.cpu().numpy(), host[k]), k
        assert np.array_equal(out[k].numpy(), t.cpu().numpy()), k
        assert int((t.long() + 1).sum().item()) == int((out[k].numpy().astype(np.int64) + 1).sum())   # a torch kernel reads it
    want = O.call_batch(t_or, docs, pairs, 48, True, True)
    ids = out["input_ids"].numpy()
    assert all(ids[i].tolist() == want[0][i] for i in range(0, len(docs), 13))
    assert np.array_equal(out["status"], np.asarray(want[4]))
    if pairs:
        tt, pl = out["token_type_ids"].numpy(), np.asarray(host["pair_len"]).reshape(-1, 2)
        for i in range(0, len(docs), 17):
            if want[4][i] == 0:
                assert tt[i, :pl[i, 1]].tolist() == want[2][i]
    del t
# a tensor made from the capsule outlives the DeviceArray object
out = tok.encode_to_device(docs[:64], max_len=16)
keep = torch.from_dlpack(out["input_ids"])
ref = out["input_ids"].numpy()
del out
gc.collect()
assert np.array_equal(keep.cpu().numpy(), ref)
print("HANDOFF OK")
