"""world_size-2 `gloo` coverage of the multi-GPU model on CPU: shard planning and the gatherv
contract of `gz_gather_rows`, with rows produced by the oracle standing in for the per-rank GPU work."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_plan_shards_balanced_and_ordered():
    from genz_tokenize.distributed import plan_shards
    rng = np.random.default_rng(0)
    sizes = rng.integers(0, 3000, size=10_000)
    offs = np.concatenate([[7], 7 + np.cumsum(sizes)]).astype(np.int64)
    for world in (1, 2, 3, 8):
        sh = plan_shards(offs, world)
        assert sh[0][0] == 0 and sh[-1][1] == 10_000
        assert all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
        b = [int(offs[hi] - offs[lo]) for lo, hi in sh]
        assert max(b) - min(b) <= 2 * 3000
    # degenerate inputs
    assert plan_shards(np.array([0, 0, 0, 0], dtype=np.int64), 2) == [(0, 0), (0, 3)] or \
        sum(hi - lo for lo, hi in plan_shards(np.array([0, 0, 0, 0], dtype=np.int64), 2)) == 3
    assert plan_shards(np.array([0], dtype=np.int64), 4) == [(0, 0)] * 4


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
    import torch.distributed as dist
    import corpus
    import gz_oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from genz_tokenize.distributed import plan_shards
    from gloo_transport import GlooTransport
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        t = O.Tables(open(corpus.VOCAB_PATH, "rb").read(), open(corpus.BPE_PATH, "rb").read())
        text, offs, _ = corpus.config_corpus(2, n_docs=301, seed=9)
        L = 24
        lo, hi = plan_shards(offs, world)[rank]
        raw = text.tobytes()
        rows = np.array([O.call(t, raw[offs[i]:offs[i + 1]].decode(), max_len=L)["input_ids"] for i in range(lo, hi)],
                        dtype=np.int32).reshape(-1, L)
        rpr = [h - l for l, h in plan_shards(offs, world)]
        got = GlooTransport(rank, world).gather_rows(rows, rpr, L, root=0)
        if rank == 0:
            want = np.array([O.call(t, raw[offs[i]:offs[i + 1]].decode(), max_len=L)["input_ids"] for i in range(301)],
                            dtype=np.int32)
            q.put(bool(np.array_equal(got, want)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gather_rows_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_ownership_and_root_bookkeeping():
    """The root-side index arithmetic of the exchange step (bench.py uses exactly these helpers): who owns which shard,
    which shard arrives from which peer in which round, block sizes in int32 words and their offsets, buffer growth."""
    from genz_tokenize.distributed import rank_shards, global_shard_id, csr_words, GatherRound
    n_shards = 8
    for world in (1, 2, 4, 8):
        owned = [rank_shards(r, world, n_shards) for r in range(world)]
        assert sorted(s for o in owned for s in o) == list(range(n_shards))           # a partition, in order
        assert all(len(o) == n_shards // world for o in owned)
        assert all(o == list(range(o[0], o[0] + len(o))) for o in owned)              # contiguous
        m = n_shards // world
        seen = set()
        for j in range(m):
            for q in range(world):
                gid = global_shard_id(q, j, world, n_shards)
                assert gid == owned[q][j]
                seen.add(gid)
        assert seen == set(range(n_shards))
    with pytest.raises(ValueError):
        rank_shards(8, 8, 8)
    with pytest.raises(ValueError):
        rank_shards(0, 9, 8)
    with pytest.raises(ValueError):
        global_shard_id(0, 1, 8, 8)
    assert [csr_words(t, 16) for t in (0, 1, 2, 3, 4, 5)] == [0, 1, 1, 2, 2, 3]
    assert [csr_words(t, 32) for t in (0, 1, 2)] == [0, 1, 2]
    from genz_tokenize.distributed import block_words
    assert block_words(5, 7, 16) == 2 * 5 + 4 and block_words(5, 7, 32) == 2 * 5 + 7 and block_words(0, 0, 16) == 0
    rng = np.random.default_rng(3)
    for world in (2, 4, 8):
        for bits in (16, 32):
            # ranks need not own equally many rows: a block is [n_real[rows] | first[rows] | entries], and where a peer's block
            # starts in the receive buffer depends on the rows of every rank before it
            rows = [int(x) for x in rng.integers(0, 3_000, size=world)]
            plan = GatherRound(world, bits, rows)
            assert plan.worst_case_words(64) == sum(block_words(n, n * 64, bits) for n in rows)
            cap_seen = 0
            for _ in range(5):                                   # several steps: the buffer only ever grows
                totals = [int(x) for x in rng.integers(0, 50_000, size=world)]
                cap = plan.announce(totals)
                assert cap >= sum(plan.words) and cap >= cap_seen
                if cap != plan.capacity:
                    plan.capacity = cap                          # (the caller re-allocates)
                cap_seen = plan.capacity
                assert plan.words == [2 * n + csr_words(t, bits) for n, t in zip(rows, totals)]
                offs = [plan.word_offset(q) for q in range(world)]
                assert offs[0] == 0 and all(offs[q + 1] - offs[q] == plan.words[q] for q in range(world - 1))
                assert offs[-1] + plan.words[-1] <= plan.capacity
                # a receive buffer laid out by the plan, read back peer by peer: each block's header is its own rows' lengths
                buf = np.full(plan.capacity, -7, dtype=np.int64)
                for q in range(world):
                    buf[plan.word_offset(q):plan.word_offset(q) + rows[q]] = q          # (the n_real part of rank q's block)
                for q in range(world):
                    o = plan.word_offset(q)
                    assert (buf[o:o + rows[q]] == q).all() and (rows[q] == 0 or buf[o + rows[q]] == -7)
            with pytest.raises(ValueError):
                GatherRound(world, bits, rows[:-1])
            with pytest.raises(ValueError):
                plan.announce([1] * (world + 1))


def _size_worker(rank, world, port, q):
    """The per-step size exchange of bench.py's exchange step: ONE all_gather_into_tensor of an int64 over gloo."""
    import time
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        size_in, size_out = torch.zeros(1, dtype=torch.int64), torch.zeros(world, dtype=torch.int64)
        ok = True
        for k in range(20):                                       # warm-up, and the values are what every rank announced
            size_in[0] = 1000 * k + rank
            dist.all_gather_into_tensor(size_out, size_in)
            ok = ok and size_out.tolist() == [1000 * k + r for r in range(world)]
        dist.barrier()
        t0 = time.perf_counter()
        reps = 200
        for k in range(reps):
            size_in[0] = k
            dist.all_gather_into_tensor(size_out, size_in)
        dt = (time.perf_counter() - t0) / reps
        if rank == 0:
            q.put((ok, dt * 1e6))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_size_exchange_of_the_exchange_step_over_gloo(world, capsys):
    """bench.py's exchange step tells every rank every block's size with one all_gather_into_tensor of a preallocated int64 over gloo
    (no pickling; round 4 used all_gather_object): right values at world 2 and 8, and its latency -- printed, and written to
    gpurun_out/size_exchange_world<N>.txt when that directory exists: the number DESIGN.md section 7 sets against one launch's
    kernel time (the exchange is hidden only while the host's part stays under it)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_size_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    ok, us = q.get(timeout=5)
    assert ok
    line = "gloo all_gather_into_tensor(int64), world %d, %d host cores: %.0f us per call" % (world, os.cpu_count() or 0, us)
    with capsys.disabled():
        print("\n    " + line)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        open(os.path.join(out, "size_exchange_world%d.txt" % world), "w").write(line + "\n")
    assert us < 50_000                                            # (a sanity bound, not a performance claim: 8 ranks may share 8 cores here)
