#!/usr/bin/env python3
"""bench.py -- the headline measurement of BASELINE.json: UTF-8 MB/s (+ tokens/s) tokenized on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (Tokenize.__call__ semantics: split, BPE, vocab lookup, frame, truncate,
pad, attention_mask) over one batch of synthetic documents that is ALREADY RESIDENT IN HBM, through the C ABI's
device entry point; with N > 1 every rank tokenizes its own shard (weak scaling: the per-GPU batch is fixed) and
the step ends with the RCCL gather of input_ids and attention_mask to rank 0.

Workload at N=1: BASELINE.json configs[2] -- 1 M mixed-length sentences, max_len=256, bundled vocab (the
configuration the roofline is quoted on).  With the default size the output is verified after the timed region
against SHA-256 digests of the REFERENCE's output (tests/golden/g5_hashes.json).

torch is used only for torch.distributed (rendezvous, barrier, max-over-ranks) and torch.cuda.synchronize();
the tokenizer itself never touches it.
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
# HBM bytes per step (all kernels of the pipeline) on the default workload, from the PMC passes kept in
# profiles/r01_v7_pmc_traffic.json (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs; KiB summed over the kernels of
# one step; the read side is not corrected for the gfx950 half-count of wide streaming reads)
MEASURED_TRAFFIC_DEFAULT_WORKLOAD = None     # filled in below from profiles/ when present


def _measured_traffic():
    path = os.path.join(ROOT, "profiles", "r01_v7_pmc_traffic.json")
    try:
        return json.load(open(path))["bytes_per_step"]
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(text, offs, max_len, budget_s=12.0):
    """The oracle (a from-scratch restatement of the reference's Python loop; kind "port") on a bounded prefix of
    the same workload, one thread, one call per document -- like the reference."""
    import gz_oracle as O
    from corpus import VOCAB_PATH, BPE_PATH
    t = O.Tables(open(VOCAB_PATH, "rb").read(), open(BPE_PATH, "rb").read())
    raw = text.tobytes()
    n = len(offs) - 1
    t0 = time.perf_counter()
    done = nbytes = ntok = 0
    while done < n:
        hi = min(n, done + 500)
        for i in range(done, hi):
            r = O.call(t, raw[offs[i]:offs[i + 1]].decode("utf-8"), max_len=max_len)
            ntok += sum(r["attention_mask"])
        nbytes += int(offs[hi] - offs[done])
        done = hi
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": round(nbytes / dt / 1e6, 4), "unit": "MB/s", "cores": 1, "kind": "port",
            "tokens_per_s": round(ntok / dt, 1),
            "sample": "first %d documents (%.2f MB) of the same workload, oracle/gz_oracle.py, %.1f s, "
                      "CPython %s single thread" % (done, nbytes / 1e6, dt, sys.version.split()[0])}


def cpu_baseline_c(text, offs, max_len, budget_s=10.0):
    """Secondary CPU figure: the plain-C restatement (oracle/gz_oracle.c), one thread, in blocks of 5000 documents.
    The reference itself is Python, so `cpu_baseline` stays the Python port; this one shows what a compiled scalar
    implementation of the same algorithm reaches on the same host."""
    import gz_oracle_c as OC
    from corpus import VOCAB_PATH, BPE_PATH
    co = OC.COracle(open(VOCAB_PATH, "rb").read(), open(BPE_PATH, "rb").read())
    text = np.ascontiguousarray(text)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
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
            "sample": "first %d documents (%.2f MB) of the same workload, oracle/gz_oracle.c (gcc -O2), %.1f s, "
                      "single thread" % (done, nbytes / 1e6, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=1_000_000, help="documents per GPU")
    ap.add_argument("--no-gather", action="store_true", help="(diagnostic) skip the RCCL gather at N > 1")
    ap.add_argument("--force-exchange", action="store_true",
                    help="(diagnostic) run the multi-GPU exchange step even with one rank (launch through torch.distributed.run)")
    ap.add_argument("--expand-at-root", action="store_true",
                    help="rank 0 also rebuilds dense [N, L] ids+mask of ALL ranks in every step (root-bound: 2 GB per rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-merge-only", action="store_true", help="skip the secondary run without the whole-word table")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one process per GPU)" % args.gpus)
        args.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_exchange:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import corpus
    from genz_tokenize import Tokenize, _native
    tok = Tokenize(device=local_rank)
    tok._sync_tables()
    ctx = tok._ctx

    # ---- workload (untimed): synthetic documents of BASELINE configs[2] (N=1) / configs[3] shards (N>1) ----------
    seed = 3 if world == 1 else 100 + rank
    text, offs, L = corpus.config_corpus(3, n_docs=args.docs, seed=seed)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    n = len(offs) - 1
    in_bytes = int(offs[-1])
    d_text = ctx.alloc(in_bytes + 64); ctx.h2d(d_text, text)
    d_off = ctx.alloc(8 * (n + 1)); ctx.h2d(d_off, offs)
    d_ids = ctx.alloc(4 * n * L); d_mask = ctx.alloc(4 * n * L); d_nreal = ctx.alloc(4 * n)
    flags = _native.GZ_PADDING | _native.GZ_TRUNCATION | _native.GZ_TIMING

    gather = (world > 1 or args.force_exchange) and not args.no_gather
    d_all_ids = d_all_mask = d_all_nreal = d_comp = 0
    all_comp = {"ptr": 0, "cap": 0}
    rows_per_rank = [n] * world
    # Double-buffered outputs: step k is tokenized into set k & 1 while the exchange of step k-1 (other set) is in flight
    # on the library's exchange stream.
    nset = 2 if gather else 1
    sets = [{"ids": d_ids, "mask": d_mask, "nreal": d_nreal}]
    if gather:
        sets.append({"ids": ctx.alloc(4 * n * L), "mask": ctx.alloc(4 * n * L), "nreal": ctx.alloc(4 * n)})
        uid = [ctx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(uid[0], rank, world)
        for st in sets:
            st["comp"] = ctx.alloc(4 * n * L)              # the rank's rows without their padding (worst case: all of it)
        if rank == 0:
            d_all_nreal = ctx.alloc(4 * n * world)

    kernel_ms = []
    last_totals = [None]
    xbits = 16 if tok.vocab_size() <= 65536 and max(tok._special_ids()) < 65536 else 32   # ids are < len(encoder)

    def exchange(st):
        """The exchange step of one tokenized batch: every rank sends its rows WITHOUT the padding (row lengths + the
        rows' real entries, CSR form) straight to rank 0 over its own xGMI link (grouped ncclSend/ncclRecv).  Rank 0
        ends up with the ids of all documents; dense [N, L] ids / mask blocks are rebuilt from that on demand
        (gz_expand_rows, done once after the timed region for the check below; --expand-at-root puts it in every step)."""
        total = ctx.compact_rows(st["ids"], st["nreal"], n, L, st["comp"], bits=xbits)
        t = torch.tensor([total], dtype=torch.int64, device="cuda")
        lst = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(lst, t)
        totals = [int(x.item()) for x in lst]
        if rank == 0 and sum(totals) > all_comp["cap"]:
            ctx.sync()
            if all_comp["ptr"]:
                ctx.free(all_comp["ptr"])
            all_comp["cap"] = int(sum(totals) * 1.05) + 1024
            all_comp["ptr"] = ctx.alloc(4 * all_comp["cap"])
        ctx.gather_rows(st["nreal"], n, 1, d_all_nreal if rank == 0 else 0, rows_per_rank, 0)
        # ranks' blocks are sent as int32 words: with 16-bit ids a block of `t` ids is (t + 1) // 2 words
        words = [(t_ * xbits // 8 + 3) // 4 for t_ in totals]
        ctx.gather_rows(st["comp"], words[rank], 1, all_comp["ptr"], words, 0)
        if rank == 0 and args.expand_at_root:
            w0 = 0
            for q in range(world):                                 # every rank's block starts on a word boundary
                ctx.expand_rows(all_comp["ptr"] + 4 * w0, d_all_nreal + 4 * n * q, n, L, root_dense["ids"] + 4 * n * L * q,
                                root_dense["mask"] + 4 * n * L * q, bits=xbits)
                w0 += words[q]
        last_totals[0] = totals

    root_dense = {}
    if gather and rank == 0 and args.expand_at_root:
        root_dense = {"ids": ctx.alloc(4 * n * L * world), "mask": ctx.alloc(4 * n * L * world)}

    def run_steps(k_steps, record):
        """k_steps tokenization steps; with an exchange, step k's kernels overlap the exchange of step k-1."""
        for k in range(k_steps):
            st = sets[k % nset]
            ctx.encode_device(d_text, d_off, 0, 0, n, L, flags, n * L, st["ids"], st["mask"], d_n_real=st["nreal"], h_text_off=offs)
            if gather:
                if k > 0:
                    ctx.exchange_select(1)
                    exchange(sets[(k - 1) % nset])
        if gather and k_steps > 0:
            ctx.exchange_select(0)
            exchange(sets[(k_steps - 1) % nset])
        # the steps were enqueued back to back (no host sync in between); this synchronises and reads the hipEvent
        # pairs recorded around every step's launches on the library's stream
        hist = ctx.timing_history(64)
        if record and k_steps > 0:
            kernel_ms.extend(hist[-k_steps:] if not gather else hist[-1:])

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup, False)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    last = sets[(args.steps - 1) % nset] if args.steps > 0 else sets[0]
    d_ids, d_mask, d_nreal = last["ids"], last["mask"], last["nreal"]

    # ---- secondary measurement (untimed for `value`): the same step with the whole-word table switched off, i.e.
    # every word through the merge loop
    merge_only = None
    if world == 1 and not args.no_merge_only:
        fl2 = flags | _native.GZ_NO_WORD_TABLE
        ms2 = []
        for _ in range(1 + min(args.steps, 5)):
            ctx.encode_device(d_text, d_off, 0, 0, n, L, fl2, n * L, d_ids, d_mask, d_n_real=d_nreal)
            ctx.sync()
            ms2.append(ctx.timing()[0])
        k2 = float(np.mean(ms2[1:]))
        merge_only = {"kernel_ms_avg": round(k2, 4), "MB_per_s_kernel": round(in_bytes / k2 / 1e3, 1)}
        # leave the buffers holding the table-on result for the verification below
        ctx.encode_device(d_text, d_off, 0, 0, n, L, flags, n * L, d_ids, d_mask, d_n_real=d_nreal)
        ctx.sync()

    # ---- secondary measurement: BASELINE configs[1] (10 k short sentences, max_len=128) -- the launch-latency regime
    small = None
    if world == 1 and not args.no_merge_only:
        t2, o2, L2 = corpus.config_corpus(2)
        n2 = len(o2) - 1
        d_t2 = ctx.alloc(len(t2) + 64); ctx.h2d(d_t2, t2)
        d_o2 = ctx.alloc(8 * (n2 + 1)); ctx.h2d(d_o2, o2)
        d_i2 = ctx.alloc(4 * n2 * L2); d_m2 = ctx.alloc(4 * n2 * L2); d_r2 = ctx.alloc(4 * n2)
        ms3 = []
        for k in range(13):
            t_a = time.perf_counter()
            ctx.encode_device(d_t2, d_o2, 0, 0, n2, L2, flags, n2 * L2, d_i2, d_m2, d_n_real=d_r2)
            ctx.sync()
            ms3.append(((time.perf_counter() - t_a) * 1e3, ctx.timing()[0]))
        wall = float(np.median([a for a, _ in ms3[3:]])); kern = float(np.median([b for _, b in ms3[3:]]))
        small = {"workload": "BASELINE configs[1]: %d short sentences (%.2f MB), max_len=%d" % (n2, len(t2) / 1e6, L2),
                 "ms_per_step_wall": round(wall, 4), "kernels_ms": round(kern, 4),
                 "MB_per_s": round(len(t2) / wall / 1e3, 1)}
        for q in (d_t2, d_o2, d_i2, d_m2, d_r2):
            ctx.free(q)

    # ---- after the timed region: counts, verification, CPU baseline -------------------------------------------------
    n_real = np.empty(n, dtype=np.int32); ctx.d2h(n_real, d_nreal)
    decode_info = None
    if world == 1 and rank == 0 and not args.no_merge_only:
        # secondary: batch decode of the step's real tokens, ids resident in HBM
        roff = np.zeros(n + 1, dtype=np.int64); np.cumsum(n_real, out=roff[1:])
        nt = int(roff[-1])
        d_c = ctx.alloc(4 * nt + 64); ctx.compact_rows(d_ids, d_nreal, n, L, d_c)
        d_ro = ctx.alloc(8 * (n + 1)); ctx.h2d(d_ro, roff)
        d_oo = ctx.alloc(8 * (n + 1))
        unk = tok.unk_token.encode()
        need = ctx.decode_device(d_c, d_ro, n, unk, 0, 0, d_oo)
        d_txt = ctx.alloc(need + 64)
        dts = []
        for _ in range(4):
            t_a = time.perf_counter()
            ctx.decode_device(d_c, d_ro, n, unk, d_txt, need, d_oo)
            dts.append(time.perf_counter() - t_a)
        dt = min(dts[1:])
        decode_info = {"workload": "decode_batch of the step's %d real tokens (%d rows), ids and text resident in HBM" % (nt, n),
                       "ms": round(dt * 1e3, 3), "tokens_per_s": round(nt / dt, 1), "text_MB_per_s": round(need / dt / 1e6, 1),
                       "text_bytes": int(need)}
        for q in (d_c, d_ro, d_oo, d_txt):
            ctx.free(q)
        # secondary (8(f) rank 3): the five preprocess.py filters chained over the same resident text
        d_po = ctx.alloc(in_bytes + 64); d_poo = ctx.alloc(8 * (n + 1))
        pps = {}
        for name, ops in (("all_five", [1, 2, 3, 4, 5]), ("remove_html", [1]), ("remove_emoji", [4])):
            dts = []
            for _ in range(3):
                t_a = time.perf_counter()
                kept = ctx.preprocess_device(ops, d_text, d_off, n, in_bytes, d_po, in_bytes, d_poo)
                dts.append(time.perf_counter() - t_a)
            pps[name] = {"ms": round(min(dts) * 1e3, 3), "MB_per_s": round(in_bytes / min(dts) / 1e6, 1), "bytes_out": int(kept)}
        decode_info["preprocess"] = {"workload": "preprocess.py filters over the step's %d documents (%.1f MB), text resident in HBM, "
                                                 "wall time incl. the size read-back" % (n, in_bytes / 1e6), **pps}
        ctx.free(d_po); ctx.free(d_poo)
    tokens_local = int(n_real.sum())
    tot = np.array([in_bytes, tokens_local, n], dtype=np.float64)
    if dist is not None:
        tt = torch.from_numpy(tot).cuda()
        dist.all_reduce(tt)
        tot = tt.cpu().numpy()
    total_bytes, total_tokens, total_docs = float(tot[0]), float(tot[1]), int(tot[2])

    verify = None
    if rank == 0 and not args.no_verify:
        ids = np.empty((n, L), dtype=np.int32); ctx.d2h(ids, d_ids)
        mask = np.empty((n, L), dtype=np.int32); ctx.d2h(mask, d_mask)
        g5 = json.load(open(os.path.join(ROOT, "tests", "golden", "g5_hashes.json"))).get("cfg3_1M")
        if world == 1 and g5 and n == g5["n_docs"] and L == g5["max_len"]:
            blk = g5["block"]
            ok = all(hashlib.sha256(ids[lo:lo + blk].tobytes()).hexdigest() == g5["ids_sha256"][k] and
                     hashlib.sha256(mask[lo:lo + blk].tobytes()).hexdigest() == g5["mask_sha256"][k]
                     for k, lo in enumerate(range(0, n, blk)))
            verify = "reference sha256 (tests/golden/g5_hashes.json cfg3_1M): %s" % ("match" if ok else "MISMATCH")
            if not ok:
                sys.exit("bench: output differs from the reference digests")
        else:
            import gz_oracle as O
            t = O.Tables(open(corpus.VOCAB_PATH, "rb").read(), open(corpus.BPE_PATH, "rb").read())
            raw = text.tobytes()
            for i in range(0, n, max(1, n // 200)):
                w = O.call(t, raw[offs[i]:offs[i + 1]].decode(), max_len=L)
                if ids[i].tolist() != w["input_ids"] or mask[i].tolist() != w["attention_mask"]:
                    sys.exit("bench: document %d differs from the oracle" % i)
            verify = "oracle on a 200-document stride sample: match"
        if gather:
            # the gathered CSR block of the last step: row counts of every rank, and rank 0's own rows rebuilt from it
            nr_all = np.empty(n * world, dtype=np.int32); ctx.d2h(nr_all, d_all_nreal)
            tot = last_totals[0]
            ok = all(int(nr_all[q * n:(q + 1) * n].sum()) == tot[q] for q in range(world))
            d_ci, d_cm = ctx.alloc(4 * n * L), ctx.alloc(4 * n * L)
            ctx.expand_rows(all_comp["ptr"], d_all_nreal, n, L, d_ci, d_cm, bits=xbits)
            ctx.sync()
            blk = np.empty((n, L), dtype=np.int32); ctx.d2h(blk, d_ci)
            mblk = np.empty((n, L), dtype=np.int32); ctx.d2h(mblk, d_cm)
            if not (ok and np.array_equal(blk, ids) and np.array_equal(mblk, mask)):
                sys.exit("bench: gathered block differs from the local rows / row counts")
            verify += "; gathered CSR block (%d rows, %d ids): row counts of every rank add up, rank 0's rows expand to its local [n, %d] ids+mask" % (
                n * world, sum(tot), L)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        k_ms = float(np.mean(kernel_ms))
        algo = in_bytes + 8 * (n + 1) + 4 * n * L * 2 + 4 * n           # bytes per launch (one rank)
        achieved = algo / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "UTF-8 MB/s tokenized (Tokenize.__call__ hot path: split + BPE + vocab lookup + pad/trunc + mask)",
            "value": round(total_bytes * args.steps / elapsed / 1e6, 2),
            "unit": "MB/s",
            "tokens_per_s": round(total_tokens * args.steps / elapsed, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8",                                          # bytes in, int32 ids out; integer / byte indexing only
            "data": "synthetic (unigram sampler over the bundled vocab.txt counts, corpus.py; seed %s)" % (
                "3" if world == 1 else "100+rank"),
            "config": {"workload": "BASELINE configs[2]: %d mixed-length sentences per GPU (70%% 5-30 / 25%% 31-120 / "
                                   "5%% 121-400 words), max_len=%d pad+trunc, bundled vocab" % (n, L),
                       "docs_total": total_docs, "input_bytes_total": int(total_bytes),
                       "tokens_total": int(total_tokens),
                       "sharding": "dp%d by documents; exchange = RCCL gatherv (grouped send/recv over direct xGMI links) of row lengths + unpadded ids (CSR, %d-bit entries) to rank 0, double-buffered under the next step's kernels%s" % (
                           world, xbits, "" if gather or world == 1 else " DISABLED (--no-gather)") if (world > 1 or gather) else "single GPU",
                       "inputs": "resident in HBM before the timed region"},
            "roofline": {"bound": "hbm",
                         "kernel": "the pipeline of one step, 9 launches on one stream: gz_brk, gz_classify, gz_scan32, gz_docw0, "
                                   "gz_words, gz_miss, gz_miss_wide, gz_long, gz_assemble (longest: gz_miss_kernel)",
                         "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": _measured_traffic() if (n == 1_000_000 and L == 256 and world == 1) else None,
                         "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summed over the step's kernels, "
                                           "profiles/r01_v7_pmc_traffic.json",
                         "algorithmic_bytes_per_launch": algo,
                         "kernel_ms_avg": round(k_ms, 4),
                         "timed_with": "hipEvents on the library's stream around the step's launches"},
            "verified": verify,
            "merge_loop_only": merge_only,
            "configs_1_small_batch": small,
            "next_rows": decode_info,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(text, offs, L)
            try:
                out["cpu_baseline_c"] = cpu_baseline_c(text, offs, L)
            except Exception as e:  # noqa: BLE001 -- a secondary figure must not cost the bench line
                out["cpu_baseline_c"] = {"error": str(e)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
