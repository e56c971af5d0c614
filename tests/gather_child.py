"""One rank of the two-process RCCL test of the exchange step (tests/test_gpu_parity.py::test_rccl_gather_two_ranks).

    python tests/gather_child.py <rank> <world> <rendezvous dir>

No torch: the ncclUniqueId travels through a file.  Every rank tokenizes its own contiguous share of one small corpus
twice (two chained encode calls), runs the exchange step of the FIRST call while the second call's kernels are in flight
(gz_exchange_select(1), as bench.py does) and then the exchange of the second; rank 0 checks that the gathered CSR
blocks expand to exactly the rows the oracle gives for every rank's documents and writes `ok` / `FAIL ...`.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "genz-tokenize_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402


def _wait_for(path, what, rank):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 60:
            sys.exit("rank %d: %s did not arrive within 60 s" % (rank, what))
        time.sleep(0.01)
    return open(path, "rb").read()


def _publish(path, data):
    with open(path + ".tmp", "wb") as f:
        f.write(data)
    os.rename(path + ".tmp", path)


def main():
    rank, world, rdv = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import corpus
    from genz_tokenize import Tokenize, _native
    from genz_tokenize.distributed import plan_shards
    tok = Tokenize(device=int(os.environ.get("GZ_CHILD_DEVICE", rank)))        # (GZ_CHILD_DEVICE: both ranks on one GPU -- RCCL refuses that)
    tok._sync_tables()
    ctx = tok._ctx
    idf = os.path.join(rdv, "uid")
    if rank == 0:
        uid = ctx.comm_unique_id()
        _publish(idf, uid)
    else:
        uid = _wait_for(idf, "the unique id", rank)
    ctx.comm_init(uid, rank, world)

    L = 48
    text, offs, _ = corpus.config_corpus(2, n_docs=4001, seed=77)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    shards = plan_shards(offs, world)                               # uneven document counts: a real gatherv
    rpr = [hi - lo for lo, hi in shards]
    lo, hi = shards[rank]
    n = hi - lo
    my_off = np.ascontiguousarray(offs[lo:hi + 1])
    d_text = ctx.alloc(len(text) + 64); ctx.h2d(d_text, text)       # absolute offsets into the whole text
    d_off = ctx.alloc(8 * (n + 1)); ctx.h2d(d_off, my_off)
    flags = _native.GZ_PADDING | _native.GZ_TRUNCATION
    sets = [{"ids": ctx.alloc(4 * n * L + 64), "mask": ctx.alloc(4 * n * L + 64), "nreal": ctx.alloc(4 * n + 64),
             "comp": ctx.alloc(4 * n * L + 64)} for _ in range(2)]
    tot_rows = sum(rpr)
    root = [{"nreal": ctx.alloc(4 * tot_rows + 64) if rank == 0 else 0, "comp": ctx.alloc(4 * tot_rows * L + 64) if rank == 0 else 0}
            for _ in range(2)]
    words = [None, None]

    def exchange(k):
        st = sets[k]
        total = ctx.compact_rows(st["ids"], st["nreal"], n, L, st["comp"], bits=16)
        # sizes travel through files too (tiny; the caller of the C ABI decides how: bench.py uses torch.distributed)
        _publish(os.path.join(rdv, "tot_%d_%d" % (k, rank)), str(total).encode())
        totals = [int(_wait_for(os.path.join(rdv, "tot_%d_%d" % (k, q)), "the size of rank %d" % q, rank)) for q in range(world)]
        w = [(t * 2 + 3) // 4 for t in totals]
        ctx.gather_rows(st["nreal"], n, 1, root[k]["nreal"], rpr, 0)
        ctx.gather_rows(st["comp"], w[rank], 1, root[k]["comp"], w, 0)
        words[k] = (w, totals)

    for k in range(2):
        ctx.encode_device(d_text, d_off, 0, 0, n, L, flags, n * L, sets[k]["ids"], sets[k]["mask"], d_n_real=sets[k]["nreal"],
                          h_text_off=my_off)
        if k == 1:
            ctx.exchange_select(1)                                   # the exchange of call 0 runs under call 1's kernels
            exchange(0)
    ctx.exchange_select(0)
    exchange(1)
    ctx.sync()
    # a refused call must not open an RCCL group (every rank would hang otherwise): a wrong local count is an error
    refused = False
    try:
        ctx.gather_rows(sets[0]["nreal"], n + 1, 1, root[0]["nreal"], rpr, 0)
    except _native.GzError:
        refused = True
    verdict = "ok" if refused else "FAIL a gather with a wrong local count was not refused"
    if rank == 0 and refused:
        import gz_oracle_c as OC
        co = OC.COracle(open(corpus.VOCAB_PATH, "rb").read(), open(corpus.BPE_PATH, "rb").read())
        wi, wm, _, _, row, _, _ = co.call_packed(np.ascontiguousarray(text), offs, max_len=L)
        want_i = wi[:int(row[-1])].reshape(-1, L); want_m = wm[:int(row[-1])].reshape(-1, L)
        for k in range(2):
            w, totals = words[k]
            d_i, d_m = ctx.alloc(4 * tot_rows * L + 64), ctx.alloc(4 * tot_rows * L + 64)
            r0 = w0 = 0
            for q in range(world):
                ctx.expand_rows(root[k]["comp"] + 4 * w0, root[k]["nreal"] + 4 * r0, rpr[q], L, d_i + 4 * r0 * L, d_m + 4 * r0 * L, bits=16)
                r0 += rpr[q]; w0 += w[q]
            ctx.sync()
            gi = np.empty((tot_rows, L), np.int32); gm = np.empty((tot_rows, L), np.int32)
            ctx.d2h(gi, d_i); ctx.d2h(gm, d_m)
            if not (np.array_equal(gi, want_i) and np.array_equal(gm, want_m)):
                verdict = "FAIL gathered block of call %d differs from the oracle" % k
            if sum(totals) != int(want_m.sum()):
                verdict = "FAIL token totals of call %d" % k
    _publish(os.path.join(rdv, "verdict_%d" % rank), verdict.encode())
    print("rank", rank, verdict, flush=True)


if __name__ == "__main__":
    main()
