#!/usr/bin/env python3
"""bench.py -- the headline measurement of BASELINE.json: UTF-8 MB/s (+ tokens/s) tokenized on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload at EVERY N: BASELINE.json configs[3] (SURVEY.md 8(d) cfg 4) -- one FIXED job of 10 M mixed-length sentences
= 8 shards x 1.25 M documents (shard s is drawn with seed 100 + s), max_len=256 pad+trunc, bundled vocab.  Rank r of G
owns shards [8r/G, 8(r+1)/G): STRONG scaling, the job does not grow with N, and N=1 runs all eight shards on one GPU.
A "step" is one pass of the hot path (Tokenize.__call__ semantics: split, BPE, vocab lookup, frame, truncate, pad,
attention_mask) over the whole job: every rank tokenizes each of its shards through the C ABI's device entry point
(one call = one launch of the kernel pipeline per shard; inputs and outputs RESIDENT IN HBM), and with N > 1 the rows
of every shard travel to rank 0 by the RCCL gather (CSR form: row lengths + unpadded 16-bit ids), overlapped with the
next shard's kernels.  `value` = bytes of the whole job x steps / wall time (max over ranks).

After the timed region EVERY rank checks the dense [n, 256] input_ids / attention_mask of each of its shards against
the committed per-shard digests (tests/golden/g5_hashes.json: cfg4_shard0..7), and rank 0 also expands every PEER's
gathered block and checks it against that peer's digests; any mismatch exits non-zero on all ranks.

At N=1 the line also carries: the roofline run of BASELINE configs[2] (1 M documents, verified against the
reference's own digests) with its three timings (kernels / device end-to-end incl. PCIe / Python end-to-end), the
merge-loop-only and OOV-sensitivity figures, configs[1] and configs[4], and the CPU baseline.

torch is used only for torch.distributed (rendezvous, barrier, max-over-ranks) and torch.cuda.synchronize(); the
tokenizer itself never touches it.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "genz-tokenize_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
N_SHARDS, SHARD_DOCS, SHARD_SEED0 = 8, 1_250_000, 100
PIPELINE = ("gz_classify, gz_scan32, gz_words2, gz_scan32 (misses), gz_mpre, gz_miss2 || gz_docw0 + gz_miss_wide + gz_long "
            "(side stream; gz_brk and the clears of the NEXT launch follow there, under gz_rows1), gz_rows1")


def kernel_source_sha16():
    """sha256 (first 16 hex digits) over the KERNEL sources (the .hip / .inc files and the two headers they share with the host
    side: kernels, launch shapes, table layouts): a PMC profile is only quoted for the kernels it was taken on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "genz-tokenize_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".inc")) or name in ("gz_kernels.h", "gz_common.h"):
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def host_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            parts = open(path).read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ---- corpus (untimed; worker processes are started before anything touches the GPU) ---------------------------------
def _make_shard(job):
    import corpus
    s, n_docs = job
    text, offs, L = corpus.config_corpus(4, n_docs=n_docs, seed=SHARD_SEED0 + s)
    return s, np.ascontiguousarray(text), np.ascontiguousarray(offs, dtype=np.int64), L


def make_shards(shard_ids, n_docs):
    if len(shard_ids) == 1:
        return [_make_shard((shard_ids[0], n_docs))]
    import multiprocessing as mp
    procs = min(len(shard_ids), max(1, host_cores()), 8)
    with mp.get_context("fork").Pool(procs) as pool:
        return pool.map(_make_shard, [(s, n_docs) for s in shard_ids], chunksize=1)


def _algo_bytes(in_bytes, n, L):
    """A = B_in + 8 (N+1) + 4 N L 2 + 4 N   (text, offsets, input_ids + attention_mask, n_tokens)."""
    return in_bytes + 8 * (n + 1) + 4 * n * L * 2 + 4 * n


def _hash_blocks(arr2d, block):
    """sha256 of consecutive row blocks of a C-contiguous int32 [n, L] array (hashlib releases the GIL: threads)."""
    from concurrent.futures import ThreadPoolExecutor
    n = arr2d.shape[0]
    los = list(range(0, n, block))
    with ThreadPoolExecutor(8) as ex:
        return list(ex.map(lambda lo: hashlib.sha256(arr2d[lo:lo + block]).hexdigest(), los))


def _check_dense(ids, mask, digest, what):
    """Compare a shard's dense output with its committed digests.  Returns None or an error string."""
    blk = digest["block"]
    hi, hm = _hash_blocks(ids, blk), _hash_blocks(mask, blk)
    bad = [k for k in range(len(hi)) if hi[k] != digest["ids_sha256"][k] or hm[k] != digest["mask_sha256"][k]]
    if bad or int(mask.sum(dtype=np.int64)) != digest["n_tokens"]:
        return "%s: blocks %s differ from tests/golden/g5_hashes.json" % (what, bad[:5])
    return None


# ---- CPU baseline ---------------------------------------------------------------------------------------------------
def _oracle_slice(job):
    """Worker of the all-cores figure: the Python oracle over documents [lo, hi) of the roofline workload, for `budget` s."""
    lo, hi, budget = job
    import corpus
    import gz_oracle as O
    text, offs, L = _CPU_WORK
    t = O.Tables(open(corpus.VOCAB_PATH, "rb").read(), open(corpus.BPE_PATH, "rb").read())
    raw = text[int(offs[lo]):int(offs[hi])].tobytes()
    base = int(offs[lo])
    t0 = time.perf_counter()
    nbytes = ntok = 0
    i = lo
    while i < hi and time.perf_counter() - t0 < budget:
        j = min(hi, i + 200)
        for d in range(i, j):
            r = O.call(t, raw[int(offs[d]) - base:int(offs[d + 1]) - base].decode("utf-8"), max_len=L)
            ntok += sum(r["attention_mask"])
        nbytes += int(offs[j] - offs[i])
        i = j
    return nbytes, ntok, time.perf_counter() - t0


_CPU_WORK = None


def cpu_baseline_small(budget_s=4.0):
    """the same loop on cfg 2 = BASELINE configs[1] (10 000 short sentences, max_len 128) -- the whole corpus when the
    budget allows (it takes the port ~ 0.6 s), one thread, one call per document."""
    global _CPU_WORK
    import corpus
    t2, o2, L2 = corpus.config_corpus(2)
    keep = _CPU_WORK
    _CPU_WORK = (np.ascontiguousarray(t2), np.ascontiguousarray(o2, dtype=np.int64), L2)
    n = len(o2) - 1
    nb, ntok, dt = _oracle_slice((0, n, budget_s))
    _CPU_WORK = keep
    ratio = None
    try:
        ratio = json.load(open(os.path.join(ROOT, "tests", "golden", "calibration.json"))).get("port_over_reference")
    except Exception:  # noqa: BLE001
        pass
    return {"value": round(nb / dt / 1e6, 4), "unit": "MB/s", "cores": 1, "kind": "port", "tokens_per_s": round(ntok / dt, 1),
            "sample": "%.2f of the %.2f MB of BASELINE configs[1] (%d sentences, max_len=%d), oracle/gz_oracle.py, %.2f s, one thread, one call per "
                      "document" % (nb / 1e6, int(o2[-1]) / 1e6, n, L2, dt),
            "reference_equivalent_MB_per_s": round(nb / dt / 1e6 / ratio, 4) if ratio else None}


def cpu_baseline(text, offs, max_len, budget_s=10.0):
    """The oracle (a from-scratch restatement of the reference's Python loop; kind "port") on a bounded prefix of the
    roofline workload (BASELINE configs[2]): one thread, one call per document -- like the reference -- and, as the
    generous figure, the same loop on every host core (multiprocessing, disjoint document ranges)."""
    global _CPU_WORK
    _CPU_WORK = (text, offs, max_len)
    n = len(offs) - 1
    nb, ntok, dt = _oracle_slice((0, n, budget_s))
    cal = {}
    try:
        cal = json.load(open(os.path.join(ROOT, "tests", "golden", "calibration.json")))
    except Exception:  # noqa: BLE001
        pass
    ratio = cal.get("port_over_reference")
    out = {"value": round(nb / dt / 1e6, 4), "unit": "MB/s", "cores": 1, "kind": "port",
           "tokens_per_s": round(ntok / dt, 1), "cpu_model": cpu_model(), "host_cores": host_cores(),
           "sample": "first %.2f MB of BASELINE configs[2] (same documents as `configs_2_roofline_run`), oracle/gz_oracle.py, "
                     "%.1f s, CPython %s, one thread, one call per document" % (nb / 1e6, dt, sys.version.split()[0]),
           "calibration": {"port_over_reference": ratio, "measured_on": cal.get("cpu"), "workload": cal.get("workload"),
                           "note": "build container, reference imported from /root/reference (tests/golden/calibrate.py); "
                                   "the port is faster than the reference's own loop by this factor"},
           "reference_equivalent_MB_per_s": round(nb / dt / 1e6 / ratio, 4) if ratio else None}
    # all cores: disjoint ranges, fork (the corpus is inherited), same per-process loop
    try:
        import multiprocessing as mp
        cores = max(1, min(host_cores(), 64))
        per = max(200, min(n // cores, 40_000))
        jobs = [(k * per, min(n, (k + 1) * per), budget_s * 0.8) for k in range(cores) if k * per < n]
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(len(jobs)) as pool:
            res = pool.map(_oracle_slice, jobs, chunksize=1)
        wall = max(r[2] for r in res)
        tot_b = sum(r[0] for r in res)
        out["all_cores"] = {"value": round(tot_b / wall / 1e6, 3), "unit": "MB/s", "cores": len(jobs),
                            "reference_equivalent_MB_per_s": round(tot_b / wall / 1e6 / ratio, 3) if ratio else None,
                            "sample": "%d processes x disjoint document ranges, %.2f MB in %.1f s (pool start %.1f s not counted)" % (
                                len(jobs), tot_b / 1e6, wall, time.perf_counter() - t0 - wall)}
    except Exception as e:  # noqa: BLE001 -- a secondary figure must not cost the bench line
        out["all_cores"] = {"error": str(e)}
    return out


def cpu_baseline_c(text, offs, max_len, budget_s=6.0):
    """Secondary CPU figure: the plain-C restatement (oracle/gz_oracle.c), one thread, in blocks of 5000 documents."""
    import gz_oracle_c as OC
    from corpus import VOCAB_PATH, BPE_PATH
    co = OC.COracle(open(VOCAB_PATH, "rb").read(), open(BPE_PATH, "rb").read())
    n = len(offs) - 1
    t0 = time.perf_counter()
    done = nbytes = ntok = 0
    while done < n and time.perf_counter() - t0 < budget_s:
        hi = min(n, done + 5000)
        _, mask, _, _, row, _, _ = co.call_packed(text, offs[done:hi + 1], max_len=max_len)
        ntok += int(mask[:int(row[-1])].sum())
        nbytes += int(offs[hi] - offs[done])
        done = hi
    dt = time.perf_counter() - t0
    return {"value": round(nbytes / dt / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": "port",
            "tokens_per_s": round(ntok / dt, 1),
            "sample": "first %d documents (%.2f MB) of BASELINE configs[2], oracle/gz_oracle.c (gcc -O2), %.1f s, "
                      "single thread" % (done, nbytes / 1e6, dt)}


# ---- helpers on device-resident workloads ---------------------------------------------------------------------------
class Resident:
    """A packed corpus uploaded once + its dense output buffers."""

    def __init__(self, ctx, text, offs, L):
        self.ctx, self.text, self.offs, self.L = ctx, text, offs, L
        self.n = len(offs) - 1
        self.in_bytes = int(offs[-1])
        self.d_text = ctx.alloc(self.in_bytes + 64); ctx.h2d(self.d_text, text)
        self.d_off = ctx.alloc(8 * (self.n + 1)); ctx.h2d(self.d_off, offs)
        self.d_ids = ctx.alloc(4 * self.n * L); self.d_mask = ctx.alloc(4 * self.n * L); self.d_nreal = ctx.alloc(4 * self.n)

    def encode(self, flags):
        self.ctx.encode_device(self.d_text, self.d_off, 0, 0, self.n, self.L, flags, self.n * self.L, self.d_ids, self.d_mask,
                               d_n_real=self.d_nreal, h_text_off=self.offs)

    def kernel_ms(self, flags, reps=5):
        """hipEvent time of the pipeline's launches, mean of `reps` back-to-back calls after one warm-up."""
        self.encode(flags); self.ctx.sync(); self.ctx.timing_history(1024)
        for _ in range(reps):
            self.encode(flags)
        h = self.ctx.timing_history(1024)
        return float(np.mean(h[-reps:]))

    def fetch(self):
        ids = np.empty((self.n, self.L), dtype=np.int32); self.ctx.d2h(ids, self.d_ids)
        mask = np.empty((self.n, self.L), dtype=np.int32); self.ctx.d2h(mask, self.d_mask)
        return ids, mask

    def free(self):
        for q in (self.d_text, self.d_off, self.d_ids, self.d_mask, self.d_nreal):
            self.ctx.free(q)


def check_vs_c_oracle(res, co=None, stride_blocks=1):
    """Whole-array comparison of a Resident's output with the C oracle (checker only)."""
    import gz_oracle_c as OC
    import corpus
    if co is None:
        co = OC.COracle(open(corpus.VOCAB_PATH, "rb").read(), open(corpus.BPE_PATH, "rb").read())
    ids, mask = res.fetch()
    blk = 20_000
    for lo in range(0, res.n, blk * stride_blocks):
        hi = min(res.n, lo + blk)
        wi, wm, _, _, row, _, _ = co.call_packed(res.text, res.offs[lo:hi + 1], max_len=res.L)
        k = int(row[-1])
        if not (np.array_equal(wi[:k], ids[lo:hi].reshape(-1)) and np.array_equal(wm[:k], mask[lo:hi].reshape(-1))):
            return False
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--docs", type=int, default=N_SHARDS * SHARD_DOCS, help="documents of the whole job (8 equal shards)")
    ap.add_argument("--no-gather", action="store_true", help="(diagnostic) skip the RCCL gather at N > 1")
    ap.add_argument("--transport", choices=("rccl", "gloo"), default="rccl",
                    help="exchange transport: rccl = gz_gather_rows (grouped ncclSend/ncclRecv over xGMI, the product path); gloo = the SAME "
                         "exchange step with the blocks carried D2H -> torch.distributed gloo send/recv -> H2D (rehearsal of the N > 1 code on "
                         "a box where RCCL cannot run, e.g. several ranks on one GPU)")
    ap.add_argument("--device", type=int, default=None, help="HIP device of this rank (default: LOCAL_RANK); --device 0 puts every rank on one GPU")
    ap.add_argument("--force-exchange", action="store_true",
                    help="(diagnostic) run the multi-GPU exchange step even with one rank (launch through torch.distributed.run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the N=1 secondary measurements")
    ap.add_argument("--switches", default="", help='(diagnostic) library switches for this run, "key=value,...": gz_debug_set through gz_switches.py')
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one process per GPU)" % args.gpus)
        args.gpus = world
    if world > N_SHARDS:
        sys.exit("bench.py: the job has %d shards; at most %d ranks" % (N_SHARDS, N_SHARDS))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    shard_docs = args.docs // N_SHARDS
    full_size = shard_docs == SHARD_DOCS

    # ---- CPU baseline (N=1 only): timed first, before the GPU is touched -- its all-cores figure forks worker processes
    cpu = {}
    cfg2 = None
    if world == 1 and not args.no_secondary:
        import corpus as _corpus
        t2_, o2_, L2_ = _corpus.config_corpus(3)
        cfg2 = (np.ascontiguousarray(t2_), np.ascontiguousarray(o2_, dtype=np.int64), L2_)
        if not args.no_cpu_baseline:
            cpu["cpu_baseline"] = cpu_baseline(*cfg2)
            try:
                cpu["cpu_baseline_configs_1"] = cpu_baseline_small()
            except Exception as e:  # noqa: BLE001 -- a secondary figure must not cost the bench line
                cpu["cpu_baseline_configs_1"] = {"error": str(e)}
            try:
                cpu["cpu_baseline_c"] = cpu_baseline_c(*cfg2)
            except Exception as e:  # noqa: BLE001 -- a secondary figure must not cost the bench line
                cpu["cpu_baseline_c"] = {"error": str(e)}

    # ---- workload (untimed): this rank's shards of the fixed job, generated before the GPU is touched ----------------
    sys.path.insert(0, os.path.join(ROOT, "genz-tokenize_amd"))
    from genz_tokenize.distributed import rank_shards as _rank_shards
    my_ids = _rank_shards(rank, world, N_SHARDS)
    t_gen = time.perf_counter()
    made = make_shards(my_ids, shard_docs)
    t_gen = time.perf_counter() - t_gen
    L = made[0][3]

    import torch
    device = local_rank if args.device is None else args.device
    torch.cuda.set_device(device)
    gloo = args.transport == "gloo"
    dist = None
    if world > 1 or args.force_exchange:
        import torch.distributed as dist
        if world == 1:                                           # --force-exchange without a launcher: a one-rank group
            for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29531"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)
        # torch.distributed carries the CONTROL plane only -- rendezvous, barriers, the few scalars of the exchange (block sizes,
        # the elapsed time) -- over gloo on the host.  The DATA plane is the library's own RCCL communicator (gz_comm_init /
        # gz_gather_rows: grouped ncclSend / ncclRecv over xGMI): RCCL has ONE user per process, and no control scalar
        # ever waits on a GPU stream.
        dist.init_process_group("gloo")
    cdev = "cpu"                                                 # where the control scalars live

    import corpus
    from genz_tokenize import Tokenize, _native
    from genz_tokenize.distributed import rank_shards, global_shard_id, GatherRound, csr_words, block_words
    import gz_switches
    switched = gz_switches.apply(args.switches) if args.switches else gz_switches.apply()      # (before any context exists)
    tok = Tokenize(device=device)
    tok._sync_tables()
    ctx = tok._ctx
    flags = _native.GZ_PADDING | _native.GZ_TRUNCATION | _native.GZ_TIMING
    xbits = 16 if tok.vocab_size() <= 65536 and max(tok._special_ids()) < 65536 else 32   # ids are < len(encoder)

    gather = (world > 1 or args.force_exchange) and not args.no_gather
    LOOK = 2                                                     # encode calls enqueued ahead of the exchange step that is being issued
    shards = []
    for s, text, offs, _ in made:
        sh = {"id": s, "text": text, "offs": offs, "n": len(offs) - 1, "in_bytes": int(offs[-1])}
        sh["d_text"] = ctx.alloc(sh["in_bytes"] + 64); ctx.h2d(sh["d_text"], text)
        sh["d_off"] = ctx.alloc(8 * (sh["n"] + 1)); ctx.h2d(sh["d_off"], offs)
        # every shard keeps its own outputs in HBM (the job's result: 20.5 GB of ids + mask at N=1).  With an exchange, the block of
        # call c leaves while calls c + 1 .. c + LOOK are already enqueued: a shard that comes round again within LOOK + 1 calls needs
        # that many output sets (one shard per rank: 3 sets; two: 2)
        nset = -(-(LOOK + 1) // len(made)) if gather else 1
        sh["sets"] = [{"ids": ctx.alloc(4 * sh["n"] * L), "mask": ctx.alloc(4 * sh["n"] * L), "nreal": ctx.alloc(4 * sh["n"])}
                      for _ in range(nset)]
        if gather:
            for st in sh["sets"]:
                # this rank's message of the exchange step: [n_real | first | the rows' real entries] (worst case: every row full)
                st["block"] = ctx.alloc(4 * block_words(sh["n"], sh["n"] * L, xbits))
        shards.append(sh)
    m = len(shards)                                               # shards per rank (the same on every rank: 8 / G)
    n = shards[0]["n"]

    # root side of the exchange: round j gathers local shard j of every rank (global shard global_shard_id(q, j) comes from
    # rank q); the bookkeeping -- block sizes and offsets -- is genz_tokenize.distributed.GatherRound.  The receive buffers
    # are sized ONCE, here, for the worst case (every row of every peer full): nothing is allocated, freed or synchronised
    # for them inside the timed region.
    # (the rows of every rank's shard of every round: shards of this job are equally large, but the bookkeeping does not rely on it)
    if dist is not None and world > 1:
        mine = torch.tensor([sh["n"] for sh in shards], dtype=torch.int64)
        every = torch.zeros(world * m, dtype=torch.int64)
        dist.all_gather_into_tensor(every, mine)
        rows_of = every.view(world, m).tolist()
    else:
        rows_of = [[sh["n"] for sh in shards]]
    rounds = [{"recv": 0, "plan": GatherRound(world, xbits, [rows_of[q][j] for q in range(world)])} for j in range(m)]
    if gather:
        if not gloo:
            uid = [ctx.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            ctx.comm_init(uid[0], rank, world)
        if rank == 0:
            for r in rounds:
                r["plan"].capacity = r["plan"].worst_case_words(L)
                r["recv"] = ctx.alloc(4 * r["plan"].capacity)

    def gloo_gatherv(d_src, my_words, d_dst, words):
        """The gatherv of gz_gather_rows over gloo: this rank's `my_words` int32 words leave the device, travel by
        torch.distributed send/recv, and land in the root's device buffer in rank order."""
        mine = np.empty(my_words, dtype=np.int32)
        if my_words:
            ctx.d2h(mine, d_src)
        if rank != 0:
            if my_words:
                dist.send(torch.from_numpy(mine), dst=0)
            return
        w0 = 0
        for q in range(world):
            k = int(words[q])
            if k:
                if q == 0:
                    buf = mine
                else:
                    t_ = torch.empty(k, dtype=torch.int32)
                    dist.recv(t_, src=q)
                    buf = t_.numpy()
                ctx.h2d(d_dst + 4 * w0, buf)
            w0 += k

    # what one exchange step costs the HOST, per step (seconds; the timed region appends): the blocking wait for THIS shard's
    # compact kernel, the size exchange over gloo, the enqueue of the gather -- and, on the device, the gather itself (events on the
    # exchange stream: gz_exchange_timing_history, read after the timed region)
    xt = {"compact_sync": [], "size_exchange": [], "gather_host": [], "block_bytes": []}
    x_on = [False]
    size_in = torch.zeros(1, dtype=torch.int64) if dist is not None else None
    size_out = torch.zeros(world, dtype=torch.int64) if dist is not None else None

    def exchange(j, st, back=None):
        """Exchange step of local shard j: every rank sends the shard's rows WITHOUT the padding -- ONE block [row lengths | first
        entries | the rows' real entries], written by the shard's own row kernel (gz_encode_emit_block) -- straight to rank 0 over
        its own xGMI link (one grouped ncclSend / ncclRecv per shard).  The host waits for the kernels of THIS shard only (the next
        shards' kernels are already queued on the main stream) to learn the block's size, then the ranks tell each other their
        block sizes over gloo (ONE all-gather into a preallocated int64 tensor: no pickling)."""
        r = rounds[j]
        plan = r["plan"]
        t_a = time.perf_counter()
        total = ctx.block_total(LOOK if back is None else back)
        t_b = time.perf_counter()
        size_in[0] = int(total)
        dist.all_gather_into_tensor(size_out, size_in)
        t_c = time.perf_counter()
        need = plan.announce(size_out.tolist())                   # entries per rank -> int32 words of every rank's whole block
        words = plan.words
        if rank == 0 and need != plan.capacity:
            sys.exit("bench: receive buffer too small (cannot happen: it holds the worst case)")
        if gloo:
            gloo_gatherv(st["block"], words[rank], r["recv"], words)
        else:
            ctx.gather_rows(st["block"], words[rank], 1, r["recv"] if rank == 0 else 0, words, 0)
        t_d = time.perf_counter()
        r["words"] = list(words)
        if x_on[0]:
            xt["compact_sync"].append(t_b - t_a); xt["size_exchange"].append(t_c - t_b); xt["gather_host"].append(t_d - t_c)
            xt["block_bytes"].append(4 * words[rank])

    kernel_ms = []
    step_no = [0]

    def run_steps(k_steps, record):
        """k_steps passes over the job; with an exchange, the block of call c leaves when calls c + 1 .. c + LOOK have been enqueued:
        the GPU always has a launch queued while the host does the exchange's control part (a blocking wait for the compact
        kernel's total, the size exchange over gloo, the gather's enqueue)."""
        import collections
        waiting = collections.deque()
        for _ in range(k_steps):
            for j, sh in enumerate(shards):
                st = sh["sets"][step_no[0] % len(sh["sets"])]
                if gather:
                    ctx.encode_emit_block(st["block"], xbits)      # the call's row kernel also writes the shard's block
                ctx.encode_device(sh["d_text"], sh["d_off"], 0, 0, sh["n"], L, flags, sh["n"] * L, st["ids"], st["mask"],
                                  d_n_real=st["nreal"], h_text_off=sh["offs"])
                if gather:
                    waiting.append((j, st))
                    if len(waiting) > LOOK:
                        ctx.exchange_select(LOOK)
                        exchange(*waiting.popleft())
            step_no[0] += 1
        while waiting:
            ctx.exchange_select(len(waiting) - 1)
            exchange(*waiting.popleft(), back=len(waiting))      # (after the pop: this many calls came behind the one whose block leaves)
        # the launches were enqueued back to back (no host sync between them unless the exchange needs one); this
        # synchronises and reads the hipEvent pairs recorded around every call's launches on the library's stream
        hist = ctx.timing_history(1024)
        if record:
            kernel_ms.extend(hist[-k_steps * m:])

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup, False)
    fence()
    if gather and not gloo:
        ctx.exchange_timing_history()                                # (forget the warm-up's gathers)
    x_on[0] = True
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    fence()
    elapsed = time.perf_counter() - t0
    x_on[0] = False
    xinfo = None
    if gather:
        # the exchange step, taken apart (per step = per shard; mean over this rank's steps, then the MAX over the ranks)
        dev_ms = ctx.exchange_timing_history() if not gloo else []
        mine = [float(np.mean(xt[k])) * 1e3 if xt[k] else 0.0 for k in ("compact_sync", "size_exchange", "gather_host")]
        mine += [float(np.mean(dev_ms)) if dev_ms else 0.0, float(np.max(dev_ms)) if dev_ms else 0.0, float(np.mean(xt["block_bytes"])) if xt["block_bytes"] else 0.0]
        tx = torch.tensor(mine, dtype=torch.float64)
        dist.all_reduce(tx, op=dist.ReduceOp.MAX)
        xinfo = [float(v) for v in tx.tolist()]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    last_set = (step_no[0] - 1) % len(shards[0]["sets"])

    # ---- after the timed region: verification of EVERY shard on EVERY rank ----------------------------------------------
    g5 = json.load(open(os.path.join(ROOT, "tests", "golden", "g5_hashes.json")))
    errors, checked = [], []
    tokens_local = 0
    co = None
    for sh in shards:
        st = sh["sets"][last_set]
        nr = np.empty(sh["n"], dtype=np.int32); ctx.d2h(nr, st["nreal"])
        tokens_local += int(nr.sum(dtype=np.int64))
        if args.no_verify:
            continue
        ids = np.empty((sh["n"], L), dtype=np.int32); ctx.d2h(ids, st["ids"])
        mask = np.empty((sh["n"], L), dtype=np.int32); ctx.d2h(mask, st["mask"])
        dg = g5.get("cfg4_shard%d" % sh["id"])
        if full_size and dg and dg["n_docs"] == sh["n"] and dg["max_len"] == L:
            e = _check_dense(ids, mask, dg, "rank %d shard %d" % (rank, sh["id"]))
            checked.append("shard %d: digests" % sh["id"])
        else:                                                     # reduced --docs: whole arrays against the C oracle
            import gz_oracle_c as OC
            co = co or OC.COracle(open(corpus.VOCAB_PATH, "rb").read(), open(corpus.BPE_PATH, "rb").read())
            wi, wm, _, _, row, _, _ = co.call_packed(sh["text"], sh["offs"], max_len=L)
            k = int(row[-1])
            e = None if (np.array_equal(wi[:k], ids.reshape(-1)) and np.array_equal(wm[:k], mask.reshape(-1))) else \
                "rank %d shard %d differs from the C oracle" % (rank, sh["id"])
            checked.append("shard %d: C oracle, whole arrays" % sh["id"])
        if int(mask.sum(dtype=np.int64)) != int(nr.sum(dtype=np.int64)):
            e = e or "rank %d shard %d: n_tokens != sum(attention_mask)" % (rank, sh["id"])
        if e:
            errors.append(e)
        del ids, mask
    # rank 0: every gathered block (its own and every peer's), expanded on the GPU and checked the same way
    if gather and rank == 0 and not args.no_verify:
        d_ci, d_cm = ctx.alloc(4 * n * L), ctx.alloc(4 * n * L)
        for j, r in enumerate(rounds):
            plan = r["plan"]
            for q in range(world):
                gid = global_shard_id(q, j, world, N_SHARDS)
                ctx.expand_block(r["recv"] + 4 * plan.word_offset(q), plan.rows[q], L, d_ci, d_cm, bits=xbits, total=plan.totals[q])
                ctx.sync()
                blk = np.empty((n, L), dtype=np.int32); ctx.d2h(blk, d_ci)
                mblk = np.empty((n, L), dtype=np.int32); ctx.d2h(mblk, d_cm)
                dg = g5.get("cfg4_shard%d" % gid)
                if full_size and dg and dg["n_docs"] == n:
                    e = _check_dense(blk, mblk, dg, "gathered block of rank %d (shard %d)" % (q, gid))
                else:
                    import gz_oracle_c as OC
                    co = co or OC.COracle(open(corpus.VOCAB_PATH, "rb").read(), open(corpus.BPE_PATH, "rb").read())
                    text_q, offs_q, _ = corpus.config_corpus(4, n_docs=n, seed=SHARD_SEED0 + gid)
                    wi, wm, _, _, row, _, _ = co.call_packed(np.ascontiguousarray(text_q), np.ascontiguousarray(offs_q, dtype=np.int64), max_len=L)
                    k = int(row[-1])
                    e = None if (np.array_equal(wi[:k], blk.reshape(-1)) and np.array_equal(wm[:k], mblk.reshape(-1))) else \
                        "gathered block of rank %d (shard %d) differs from the C oracle" % (q, gid)
                if int(mblk.sum(dtype=np.int64)) != plan.totals[q]:
                    e = e or "gathered block of rank %d: ids received != ids announced" % q
                if e:
                    errors.append(e)
                checked.append("gathered shard %d (from rank %d)" % (gid, q))
                del blk, mblk
        ctx.free(d_ci); ctx.free(d_cm)

    tot = np.array([sum(sh["in_bytes"] for sh in shards), tokens_local, sum(sh["n"] for sh in shards), len(errors)], dtype=np.float64)
    if dist is not None:
        tt = torch.from_numpy(tot).to(cdev)
        dist.all_reduce(tt)
        tot = tt.cpu().numpy()
    total_bytes, total_tokens, total_docs, n_err = float(tot[0]), float(tot[1]), int(tot[2]), int(tot[3])
    if n_err:
        for e in errors:
            print("bench: VERIFICATION FAILED: " + e, file=sys.stderr, flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(1)
    verify = None if args.no_verify else (
        "every rank: dense input_ids + attention_mask of each of its shards == %s" % (
            "committed per-shard sha256 digests (tests/golden/g5_hashes.json cfg4_shard*: C oracle over the whole shard, "
            "the reference itself over the first 20 000 documents)" if full_size else "C oracle, whole arrays")
        + ("; rank 0: the gathered CSR block of EVERY rank expanded to dense rows and checked the same way (%d blocks)" % (m * world) if gather else ""))

    # ---- N=1 secondary measurements (untimed for `value`) --------------------------------------------------------------
    sec = {}
    if world == 1 and rank == 0 and not args.no_secondary:
        # free the job's buffers first (20+ GB); the secondary workloads are small
        for sh in shards:
            for q in [sh["d_text"], sh["d_off"]] + [x for st in sh["sets"] for x in (st["ids"], st["mask"], st["nreal"])]:
                ctx.free(q)
        sec = {"headline_host_paths": headline_e2e(ctx, shards, L, int(total_tokens))}
        sec.update(secondary(ctx, tok, flags, args, cfg2))
        sec.update(cpu)
        if "cpu_baseline_configs_1" in sec and "configs_1_small_batch" in sec:       # (the CPU figure beside its workload)
            sec["configs_1_small_batch"]["cpu_baseline"] = sec.pop("cpu_baseline_configs_1")

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        k_ms = float(np.mean(kernel_ms))                                  # one launch of the pipeline = one shard
        in_b = shards[0]["in_bytes"] if m == 1 else float(np.mean([sh["in_bytes"] for sh in shards]))
        algo = _algo_bytes(in_b, n, L)
        achieved = algo / (k_ms * 1e-3) / 1e9
        shard_traffic = _pmc_traffic("r06_pmc_traffic_shard.json")
        out = {
            "metric": "UTF-8 MB/s tokenized (Tokenize.__call__ hot path: split + BPE + vocab lookup + pad/trunc + mask)",
            "value": round(total_bytes * args.steps / elapsed / 1e6, 2),
            "unit": "MB/s",
            "tokens_per_s": round(total_tokens * args.steps / elapsed, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "value_is": "timing (i) of SURVEY.md 8(d): the whole job through the kernels, inputs resident in HBM when the clock starts, outputs left "
                        "in HBM (wall clock over the steps, max over ranks).  The PCIe-inclusive rate of the same job -- timing (ii), host "
                        "buffers in and out -- is headline_host_paths.MB_per_s; timing (iii), Python lists of str, is "
                        "configs_2_roofline_run.timings.python_e2e_ms.",
            "dtype": "u8",                                          # bytes in, int32 ids out; integer / byte indexing only
            "data": "synthetic (unigram sampler over the bundled vocab.txt counts, corpus.py; shard s = seed %d + s; generated in %.1f s, untimed)" % (SHARD_SEED0, t_gen),
            "config": {"workload": "BASELINE configs[3]: ONE fixed job of %d mixed-length sentences (70%% 5-30 / 25%% 31-120 / 5%% 121-400 "
                                   "words) = %d shards x %d documents, max_len=%d pad+trunc, bundled vocab; rank r of %d owns shards "
                                   "[%dr/%d, %d(r+1)/%d); a step = the whole job once" % (
                                       total_docs, N_SHARDS, n, L, world, N_SHARDS, world, N_SHARDS, world),
                       "docs_total": total_docs, "input_bytes_total": int(total_bytes), "tokens_total": int(total_tokens),
                       "shards_per_rank": m, "launches_per_step_per_rank": m,
                       "sharding": ("dp%d by documents (contiguous shards); exchange = %s "
                                    "of row lengths + unpadded ids (CSR, %d-bit entries) to rank 0, overlapped with the next shard's kernels%s" % (
                                        world, "gatherv over torch.distributed gloo (D2H -> send/recv -> H2D: the REHEARSAL transport, not the product's)" if gloo
                                        else "RCCL gatherv (grouped send/recv over direct xGMI links)", xbits,
                                        "" if gather else " DISABLED (--no-gather)")) if world > 1 or gather else "single GPU: all 8 shards, one after the other",
                       "inputs": "resident in HBM before the timed region; outputs of every shard stay resident"},
            "roofline": {"bound": "hbm",
                         "kernel": "one launch of the pipeline = one shard of %d documents, on one stream: %s" % (n, PIPELINE),
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": shard_traffic["bytes_per_step"] if shard_traffic else None,
                         # the text is streamed twice with 16-byte-per-lane loads (classify, words), which FETCH_SIZE counts at half
                         # their bytes on gfx950 (MI355X_MICROARCH.md, HBM): the corrected figure adds the uncounted half of both passes
                         "traffic_corrected": int(shard_traffic["bytes_per_step"] + in_b) if shard_traffic else None,
                         "traffic_note": ("profiles/r06_pmc_traffic_shard.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over one "
                                          "launch on shard 0 (1.25 M documents), same kernel sources (sha %s); reads of 16-B streams are "
                                          "half-counted on gfx950, not corrected" % kernel_source_sha16()) if shard_traffic else
                                         "null: profiles/r06_pmc_traffic_shard.json is absent or was taken on other kernel sources",
                         "algorithmic_bytes_per_launch": int(algo),
                         "kernel_ms_avg": round(k_ms, 4), "launches_timed": len(kernel_ms),
                         "timed_with": "hipEvents on the library's stream around each launch of the pipeline, inside the timed region"},
            "verified": verify, "verified_items": checked if len(checked) <= 24 else checked[:24] + ["..."],
        }
        if switched:
            out["switches"] = dict(switched)                         # (diagnostic run: gz_debug_set pairs in force)
        if xinfo is not None:
            host_ms = xinfo[1] + xinfo[2]                            # the host's own part: size exchange + the gather's enqueue
            out["exchange"] = {
                "what": "one exchange step = one shard's block [int32 row lengths | uint32 first entries | %d-bit real entries] -- made by the "
                        "shard's own encode call, behind its kernels (gz_encode_emit_block) -- from every peer to rank 0; per step, "
                        "mean over a rank's steps inside the timed region, MAX over the ranks" % xbits,
                "transport": "gloo rehearsal (D2H -> send/recv -> H2D)" if gloo else "RCCL grouped ncclSend/ncclRecv",
                "bytes_per_peer": int(xinfo[5]),
                "compact_sync_ms": round(xinfo[0], 4),          # gz_block_total: the host waits until THIS shard's kernels and its block are done (two calls are queued behind it: in
                                                                # the steady state this is the GPU's pace, one launch period -- not host work)
                "size_exchange_ms": round(xinfo[1], 4),         # one gloo all_gather_into_tensor of an int64
                "gather_enqueue_ms": round(xinfo[2], 4),        # host time of gz_gather_rows (gloo rehearsal: the whole transfer)
                "gather_enqueue_to_done_ms": None if gloo else round(xinfo[3], 4),     # hipEvents on the exchange stream
                "gather_enqueue_to_done_ms_max": None if gloo else round(xinfo[4], 4),
                "link_GB_per_s_per_peer": None if (gloo or xinfo[3] <= 0) else round(xinfo[5] / (xinfo[3] * 1e-3) / 1e9, 2),
                "host_control_ms": round(host_ms, 4),
                "kernels_ms_per_step": round(k_ms, 4),
                # the exchange of shard k runs under the kernels of shards k + 1, k + 2: it is hidden while both the host's own part and
                # the transfer stay under one launch's kernels
                "hidden_under_kernels": bool(host_ms < k_ms and (gloo or xinfo[3] < k_ms)),
            }
        out.update(sec)
        if "headline_host_paths" in sec:
            # timing (ii) of SURVEY.md 8(d) for the SAME job, first-class: pinned host text in, pinned CSR rows out (PCIe-bound)
            out["value_host_e2e_MB_per_s"] = sec["headline_host_paths"].get("MB_per_s")
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _pmc_traffic(name):
    """A committed PMC traffic summary, or None unless it was taken on this pipeline and these kernel sources."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", name)))
        if tj.get("pipeline") == PIPELINE and tj.get("source_sha16") == kernel_source_sha16():
            return tj
    except Exception:  # noqa: BLE001
        pass
    return None


def headline_e2e(ctx, shards, L, n_tok_expected):
    """SURVEY.md 8(d) timing (ii) for the HEADLINE job (BASELINE configs[3], all 8 shards on this GPU): host buffers in, host
    buffers out.  Every shard goes through gz_encode_batch_csr from pinned host memory: its text travels H2D in 32-MB
    sub-batches under the kernels of the sub-batch before, and the rows' real entries (16-bit) + 4 bytes per document
    travel back under the kernels of the sub-batch after.  The pinned buffers are allocated and filled before the clock starts."""
    nmax = max(sh["n"] for sh in shards)
    bmax = max(sh["in_bytes"] for sh in shards)
    ptext = ctx.pinned_empty(bmax, np.uint8)
    ptok = ctx.pinned_empty(min(nmax * L, bmax + 2 * nmax), np.uint16)
    pnr = ctx.pinned_empty(nmax, np.int32)
    times, tokens, pcie = [], 0, 0
    for rep in range(2):                                         # (the first pass warms the workspace allocations up)
        times, tokens, pcie = [], 0, 0
        for sh in shards:
            ptext[:sh["in_bytes"]] = sh["text"]
            t_a = time.perf_counter()
            toks, nr = ctx.encode_csr(ptext[:sh["in_bytes"]], sh["offs"], L, 16, tokens=ptok, n_real=pnr)
            times.append(time.perf_counter() - t_a)
            tokens += int(len(toks))
            pcie += sh["in_bytes"] + 8 * (sh["n"] + 1) + int(toks.nbytes) + 4 * sh["n"]
    total_b = sum(sh["in_bytes"] for sh in shards)
    dt = sum(times)
    del ptext, ptok, pnr
    return {"what": "timing (ii) of SURVEY.md 8(d) for the headline job: the %d shards one after the other through gz_encode_batch_csr, pinned host "
                    "text in, pinned host CSR rows (16-bit ids + row lengths) out; H2D, kernels and D2H of consecutive 32-MB sub-batches overlap "
                    "on three streams" % len(shards),
            "device_e2e_ms": round(dt * 1e3, 3), "MB_per_s": round(total_b / dt / 1e6, 1), "bytes_over_pcie": int(pcie),
            "pcie_GB_per_s": round(pcie / dt / 1e9, 2), "tokens_match_headline": tokens == n_tok_expected}


def secondary(ctx, tok, flags, args, cfg2):
    """Everything the N=1 line reports besides the headline: the configs[2] roofline run with its three timings, the
    merge-loop-only and OOV figures, configs[1], configs[4], the "next" rows, the CPU baseline."""
    import corpus
    from genz_tokenize import Tokenize, _native
    out = {}
    g5 = json.load(open(os.path.join(ROOT, "tests", "golden", "g5_hashes.json")))
    smp = corpus.Sampler()

    # ---- BASELINE configs[2]: 1 M mixed-length sentences, the rocprof roofline workload ------------------------------
    text, offs, L = cfg2
    R = Resident(ctx, text, offs, L)
    k_ms = R.kernel_ms(flags, reps=10)
    ids, mask = R.fetch()
    dg = g5["cfg3_1M"]
    err = _check_dense(ids, mask, dg, "configs[2]")
    if err:
        sys.exit("bench: " + err)
    n_tok = int(mask.sum(dtype=np.int64))
    del ids, mask
    algo = _algo_bytes(R.in_bytes, R.n, L)
    traffic = _pmc_traffic("r06_pmc_traffic.json")
    # (ii) device end-to-end: host buffers in, host buffers out (PCIe both ways).  The library's host path for batches is
    # gz_encode_batch_csr: sub-batches, text H2D / kernels / D2H on three streams, and only the rows' real entries
    # (16-bit) + 4 bytes per document come back; the buffers are pinned (gz_host_alloc), as SURVEY.md 8(d) (ii) says.
    ptext = ctx.pinned_empty(R.in_bytes, np.uint8); ptext[:] = text
    ptok = ctx.pinned_empty(min(R.n * L, R.in_bytes + 2 * R.n), np.uint16)
    pnr = ctx.pinned_empty(R.n, np.int32)
    e2e = []
    for _ in range(4):
        t_a = time.perf_counter()
        toks, nr = ctx.encode_csr(ptext, offs, L, 16, tokens=ptok, n_real=pnr)
        e2e.append(time.perf_counter() - t_a)
    ok_e2e = len(toks) == n_tok and int(nr.sum(dtype=np.int64)) == n_tok
    csr_bytes = int(toks.nbytes + nr.nbytes)
    # ... and what the same call costs when the caller wants the dense [N, L] int32 arrays in pageable host memory
    e2e_dense = []
    for _ in range(3):                       # (the first call also makes the library's pinned staging buffers; the host's 2 GB of first-touch stores vary by box)
        t_a = time.perf_counter()
        r = ctx.encode(text, offs, None, None, L, True, True)
        e2e_dense.append(time.perf_counter() - t_a)
        ok_e2e = ok_e2e and int(r["attention_mask"].sum(dtype=np.int64)) == n_tok
        del r                                # (outside the clock: giving 2 GB of arrays back to the system takes as long as the call)
    del toks, nr, ptext, ptok, pnr
    # (iii) Python end-to-end: list of str in (packing included), numpy arrays out
    raw = text.tobytes()
    docs = [raw[offs[i]:offs[i + 1]].decode("utf-8") for i in range(R.n)]
    py_runs = []
    ok_py = True
    for _ in range(2):                       # (the first call also allocates the object's pinned text arena: both runs are in the line, the better one counts --
        t_a = time.perf_counter()            #  as for every other timing here)
        r = tok.encode_batch(docs, max_len=L)
        py_runs.append(time.perf_counter() - t_a)
        ok_py = ok_py and int(r["attention_mask"].sum(dtype=np.int64)) == n_tok
        del r
    py_e2e = min(py_runs)
    # (iii') the path a model-feed user takes: list of str in, dense [N, L] rows LEFT IN HBM (DLPack hand-off): packing + H2D of the text +
    # kernels; only n_real [N] comes back
    py_dev = []
    for _ in range(2):
        t_a = time.perf_counter()
        dv = tok.encode_to_device(docs, max_len=L)
        py_dev.append(time.perf_counter() - t_a)
        ok_py = ok_py and int(np.asarray(dv["n_real"]).sum(dtype=np.int64)) == n_tok
        del dv
    del docs, raw
    out["configs_2_roofline_run"] = {
        "workload": "BASELINE configs[2]: %d mixed-length sentences (%.1f MB), max_len=%d, one launch of the pipeline" % (R.n, R.in_bytes / 1e6, L),
        "verified": "reference sha256 (tests/golden/g5_hashes.json cfg3_1M, computed by the reference itself): match",
        "timings": {"kernels_ms": round(k_ms, 4), "device_e2e_ms": round(min(e2e) * 1e3, 3), "python_e2e_ms": round(py_e2e * 1e3, 2), "python_e2e_ms_runs": [round(x * 1e3, 2) for x in py_runs],
                    "python_to_device_ms": round(min(py_dev) * 1e3, 2),
                    "device_e2e_dense_pageable_ms": round(min(e2e_dense) * 1e3, 2), "device_e2e_dense_pageable_ms_runs": [round(x * 1e3, 2) for x in e2e_dense], "device_e2e_bytes_over_pcie": int(R.in_bytes + 8 * (R.n + 1) + csr_bytes),
                    "MB_per_s": {"kernels": round(R.in_bytes / k_ms / 1e3, 1), "device_e2e": round(R.in_bytes / min(e2e) / 1e6, 1), "device_e2e_dense_pageable": round(R.in_bytes / min(e2e_dense) / 1e6, 1),
                                 "python_e2e": round(R.in_bytes / py_e2e / 1e6, 1), "python_to_device": round(R.in_bytes / min(py_dev) / 1e6, 1)},
                    "what": "(i) hipEvents around the launches, inputs/outputs in HBM; (ii) gz_encode_batch_csr on pinned host buffers: "
                            "text + offsets H2D, kernels, D2H of n_real + the rows' real entries (16-bit), sub-batches on three streams "
                            "(device_e2e_dense_pageable_ms: gz_encode_batch returning dense [N, L] ids + mask into pageable numpy arrays); "
                            "(iii) Tokenize.encode_batch(list of str): UTF-8 packing + the dense host path; python_to_device_ms: "
                            "Tokenize.encode_to_device(list of str) -- packing, text H2D, kernels, the dense rows stay in HBM (DLPack).  Token totals of (ii)/(iii) "
                            "equal (i): %s" % (ok_e2e and ok_py)},
        "roofline": {"bound": "hbm", "achieved": round(algo / k_ms / 1e6, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(algo / k_ms / 1e6 / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_launch": int(algo),
                     "kernel_ms_avg": round(k_ms, 4),
                     "traffic": traffic["bytes_per_step"] if traffic else None,
                     "traffic_source": "profiles/r06_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, summed over the "
                                       "launch's kernels; reads not corrected for the gfx950 half-count)" if traffic else
                                       "none for this build (profiles/r06_pmc_traffic.json absent or taken on other kernel sources)"}}
    # ---- the same documents WITHOUT padding (max_len=None, the reference's default call: ragged rows + int64 row offsets): all kernels
    # of the call by hipEvents, inputs / outputs resident
    capu = R.in_bytes + 2 * R.n
    d_ui, d_um, d_uo = ctx.alloc(4 * capu), ctx.alloc(4 * capu), ctx.alloc(8 * (R.n + 1))
    uflags = _native.GZ_PADDING | _native.GZ_TRUNCATION | _native.GZ_MAX_LEN_NONE | _native.GZ_TIMING
    ums = []
    for _ in range(4):
        ctx.encode_device(R.d_text, R.d_off, 0, 0, R.n, 0, uflags, capu, d_ui, d_um, d_row_off=d_uo, d_n_real=R.d_nreal, h_text_off=R.offs)
        ctx.sync()
        ums.append(ctx.timing()[3])
    uro = np.empty(R.n + 1, dtype=np.int64); ctx.d2h(uro, d_uo)
    for q in (d_ui, d_um, d_uo):
        ctx.free(q)
    u_ms = float(np.mean(ums[1:]))
    u_algo = R.in_bytes + 8 * (R.n + 1) + 8 * int(uro[-1]) + 8 * (R.n + 1) + 4 * R.n       # text + offsets in; ids + mask + row offsets + n_real out
    out["configs_2_roofline_run"]["unpadded_run"] = {
        "what": "the same documents with max_len=None (ragged rows + int64 row offsets; a count pass, the scan, every row written once): all kernels of the call by hipEvents",
        "kernel_ms_avg": round(u_ms, 4), "MB_per_s_kernel": round(R.in_bytes / u_ms / 1e3, 1), "tokens": int(uro[-1]),
        "algorithmic_bytes": int(u_algo), "roofline_frac": round(u_algo / u_ms / 1e6 / HBM_PEAK_GBS, 5),
        "verified": "row offsets ascend, total >= the dense run's %d real tokens (rows cut at max_len there); the rows themselves: "
                    "tests/test_gpu_parity.py (whole arrays against the C oracle)" % n_tok}
    if int(uro[-1]) < n_tok or np.any(np.diff(uro) < 2):
        sys.exit("bench: the unpadded run of configs[2] yields %d tokens, the dense run %d" % (int(uro[-1]), n_tok))
    # ---- the same step with the whole-word tables off: every word through the merge loop
    k2 = R.kernel_ms(flags | _native.GZ_NO_WORD_TABLE, reps=3)
    out["merge_loop_only"] = {"kernel_ms_avg": round(k2, 4), "MB_per_s_kernel": round(R.in_bytes / k2 / 1e3, 1)}

    # ---- "next" rows: batch decode and the text pre-pass on the same resident data -------------------
    R.encode(flags); ctx.sync()
    n_real = np.empty(R.n, dtype=np.int32); ctx.d2h(n_real, R.d_nreal)
    roff = np.zeros(R.n + 1, dtype=np.int64); np.cumsum(n_real, out=roff[1:])
    nt = int(roff[-1])
    d_c = ctx.alloc(4 * nt + 64); ctx.compact_rows(R.d_ids, R.d_nreal, R.n, L, d_c)
    d_ro = ctx.alloc(8 * (R.n + 1)); ctx.h2d(d_ro, roff)
    d_oo = ctx.alloc(8 * (R.n + 1))
    unk = tok.unk_token.encode()
    need = ctx.decode_device(d_c, d_ro, R.n, unk, 0, 0, d_oo)
    d_txt = ctx.alloc(need + 64)
    dts = []
    for _ in range(4):
        t_a = time.perf_counter()
        ctx.decode_device(d_c, d_ro, R.n, unk, d_txt, need, d_oo)
        dts.append(time.perf_counter() - t_a)
    dt = min(dts[1:])
    nxt = {"decode_batch": {"workload": "the step's %d real tokens (%d rows), ids and text resident in HBM" % (nt, R.n),
                            "ms": round(dt * 1e3, 3), "tokens_per_s": round(nt / dt, 1), "text_MB_per_s": round(need / dt / 1e6, 1)}}
    for q in (d_c, d_ro, d_oo, d_txt):
        ctx.free(q)
    d_po = ctx.alloc(R.in_bytes + 64); d_poo = ctx.alloc(8 * (R.n + 1))
    pps = {}
    for name, ops in (("all_five", [1, 2, 3, 4, 5]), ("remove_html", [1])):
        dts = []
        for _ in range(3):
            t_a = time.perf_counter()
            kept = ctx.preprocess_device(ops, R.d_text, R.d_off, R.n, R.in_bytes, d_po, R.in_bytes, d_poo)
            dts.append(time.perf_counter() - t_a)
        pps[name] = {"ms": round(min(dts) * 1e3, 3), "MB_per_s": round(R.in_bytes / min(dts) / 1e6, 1), "bytes_out": int(kept)}
    nxt["preprocess"] = pps
    ctx.free(d_po); ctx.free(d_poo)
    out["next_rows"] = nxt

    R.free()

    # ---- OOV sensitivity: the headline corpus is drawn from the vocabulary itself (97 % of its words hit the load-time
    # whole-word table); here 5 % / 20 % of the words are replaced by random letters, which always run the merge loop
    n_oov = 200_000
    t0_, o0_, _ = corpus.config_corpus(3, n_docs=n_oov, sampler=smp)
    o0_ = np.ascontiguousarray(o0_, dtype=np.int64)
    oov = {"workload": "first-principles variant of configs[2]: %d documents, a fraction of the words replaced by same-length random "
                       "[a-z0-9] strings (corpus.add_typos), verified against the C oracle (whole arrays)" % n_oov}
    for rate in (0.0, 0.05, 0.20):
        t_ = np.ascontiguousarray(corpus.add_typos(t0_, o0_, seed=11, rate=rate)) if rate else np.ascontiguousarray(t0_)
        Rz = Resident(ctx, t_, o0_, L)
        kz = Rz.kernel_ms(flags, reps=5)
        ok = check_vs_c_oracle(Rz, stride_blocks=2)
        if not ok:
            sys.exit("bench: OOV run (rate %.2f) differs from the C oracle" % rate)
        oov["rate_%.2f" % rate] = {"kernel_ms": round(kz, 4), "MB_per_s_kernel": round(Rz.in_bytes / kz / 1e3, 1)}
        Rz.free()
    out["oov_sensitivity"] = oov

    # ---- BASELINE configs[1]: 10 k short sentences -- the launch-latency regime -----------------------------------------
    t2, o2, L2 = corpus.config_corpus(2, sampler=smp)
    R2 = Resident(ctx, np.ascontiguousarray(t2), np.ascontiguousarray(o2, dtype=np.int64), L2)
    ms3 = []
    for _ in range(13):
        t_a = time.perf_counter()
        R2.encode(flags); ctx.sync()
        ms3.append(((time.perf_counter() - t_a) * 1e3, ctx.timing()[0]))
    i2, m2 = R2.fetch()
    e2 = _check_dense(i2, m2, g5["cfg2_10k"], "configs[1]")
    if e2:
        sys.exit("bench: " + e2)
    wall = float(np.median([a for a, _ in ms3[3:]])); kern = float(np.median([b for _, b in ms3[3:]]))
    out["configs_1_small_batch"] = {"workload": "BASELINE configs[1]: %d short sentences (%.2f MB), max_len=%d; one launch (gz_small_kernel)"
                                                % (R2.n, R2.in_bytes / 1e6, L2),
                                    "ms_per_step_wall": round(wall, 4), "kernels_ms": round(kern, 4),
                                    "MB_per_s": round(R2.in_bytes / wall / 1e3, 1), "verified": "reference sha256 (cfg2_10k): match"}
    R2.free()

    # ---- BASELINE configs[4]: Tokenize.fromFile custom tables (100 k-entry vocab, header-less merges), 4 k-char documents
    import tempfile
    v, b = corpus.custom_tables()
    tmp = tempfile.mkdtemp()
    open(os.path.join(tmp, "v"), "wb").write(v); open(os.path.join(tmp, "b"), "wb").write(b)
    tok5 = Tokenize.fromFile(os.path.join(tmp, "v"), os.path.join(tmp, "b"))
    tok5._sync_tables()
    t5, o5, L5 = corpus.config_corpus(5, sampler=smp)
    R5 = Resident(tok5._ctx, np.ascontiguousarray(t5), np.ascontiguousarray(o5, dtype=np.int64), L5)
    k5 = R5.kernel_ms(flags, reps=5)
    i5, m5 = R5.fetch()
    d5 = g5.get("cfg5_50k")
    v5 = "not verified (cfg5_50k digests missing)"
    if d5 and d5["n_docs"] == R5.n:
        pd = dict(d5["padded"]); pd["block"] = d5["block"]
        e5 = _check_dense(i5, m5, pd, "configs[4]")
        if e5:
            sys.exit("bench: " + e5)
        v5 = "C-oracle sha256 over all %d documents (tests/golden/g5_hashes.json cfg5_50k; its first 300 documents also hashed by the reference): match" % R5.n
    a5 = _algo_bytes(R5.in_bytes, R5.n, L5)
    t5j = _pmc_traffic("r06_pmc_traffic_cfg4.json")
    # ... and the unpadded run SURVEY.md 8(d) asks for beside it (max_len=None: ragged rows + row offsets; parity: the suite's
    # test_cfg5_full_size_padded_and_unpadded): kernels by hipEvents, inputs / outputs resident
    c5 = tok5._ctx
    cap5 = R5.in_bytes + 2 * R5.n
    d_ri, d_rm, d_ro = c5.alloc(4 * cap5), c5.alloc(4 * cap5), c5.alloc(8 * (R5.n + 1))
    rag_flags = _native.GZ_PADDING | _native.GZ_TRUNCATION | _native.GZ_MAX_LEN_NONE | _native.GZ_TIMING
    rag = []
    for _ in range(4):
        c5.encode_device(R5.d_text, R5.d_off, 0, 0, R5.n, 0, rag_flags, cap5, d_ri, d_rm, d_row_off=d_ro, d_n_real=R5.d_nreal, h_text_off=R5.offs)
        c5.sync()
        rag.append(c5.timing()[3])
    ro5 = np.empty(R5.n + 1, dtype=np.int64); c5.d2h(ro5, d_ro)
    u5 = None
    if d5 and d5["n_docs"] == R5.n and "unpadded" in d5:
        u5 = int(ro5[-1]) == int(d5["unpadded"]["n_tokens"])        # (the rows themselves: tests/test_gpu_parity.py, whole digests)
    a5u = R5.in_bytes + 8 * (R5.n + 1) + 8 * int(ro5[-1]) + 8 * (R5.n + 1) + 4 * R5.n        # text + offsets in; ids + mask + row offsets + n_real out
    for q in (d_ri, d_rm, d_ro):
        c5.free(q)
    out["configs_4_long_docs"] = {
        "workload": "BASELINE configs[4]: Tokenize.fromFile custom tables (100 000-entry vocab, %d merges without the #version header), "
                    "%d documents of <= 4 000 characters (%.1f MB), max_len=%d pad+trunc" % (tok5._ctx.table_info()[2], R5.n, R5.in_bytes / 1e6, L5),
        "kernel_ms_avg": round(k5, 4), "MB_per_s_kernel": round(R5.in_bytes / k5 / 1e3, 1), "verified": v5,
        "unpadded_run": {"what": "the same documents with max_len=None (ragged rows + int64 row offsets): all kernels of the call by hipEvents",
                         "kernel_ms_avg": round(float(np.mean(rag[1:])), 4), "MB_per_s_kernel": round(R5.in_bytes / float(np.mean(rag[1:])) / 1e3, 1),
                         "tokens": int(ro5[-1]), "algorithmic_bytes": int(a5u),
                         "roofline_frac": round(a5u / float(np.mean(rag[1:])) / 1e6 / HBM_PEAK_GBS, 5),
                         "token_total_matches_digest_file": u5},
        "roofline": {"bound": "hbm", "achieved": round(a5 / k5 / 1e6, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(a5 / k5 / 1e6 / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_launch": int(a5),
                     "traffic": t5j["bytes_per_step"] if t5j else None,
                     "traffic_source": "profiles/r06_pmc_traffic_cfg4.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/prof_cfg5.py)"
                                       if t5j else "none for this build (profiles/r06_pmc_traffic_cfg4.json absent or taken on other kernel sources)"}}
    R5.free()
    return out


if __name__ == "__main__":
    main()
