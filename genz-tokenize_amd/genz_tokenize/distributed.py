"""Multi-GPU model of the hot path: one process per GPU, documents sharded, rows gathered to a root.

The reference has no parallelism of any kind; documents are independent (no word cache, no
cross-document state), so the batch shards by contiguous document ranges and the only exchange step is the
gather of the `[n_r, max_len]` int32 blocks (input_ids, attention_mask) to the root rank, done by the C ABI's
`gz_gather_rows` (grouped ncclSend/ncclRecv over xGMI: every peer uses its own link to the root).

The shard planning lives here; the transport is RCCL inside the C library (`RcclTransport`).  The CPU test-suite
exercises the same gatherv contract over `gloo` with a stand-in transport kept under tests/ (tests/gloo_transport.py).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def plan_shards(offsets: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """Contiguous document ranges [lo, hi) per rank, balanced by BYTES: rank r's range ends at the document
    boundary nearest to total_bytes*(r+1)/world.  Ranges concatenate to the input order."""
    n = len(offsets) - 1
    base, total = int(offsets[0]), int(offsets[-1] - offsets[0])
    cuts = [0]
    for r in range(1, world):
        target = base + total * r / world
        k = int(np.searchsorted(offsets, target, side="left"))
        if k > 0 and k <= n and abs(int(offsets[k - 1]) - target) <= abs(int(offsets[min(k, n)]) - target):
            k -= 1
        cuts.append(min(max(k, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


class RcclTransport:
    """Device-resident gatherv of int32 row blocks to a root rank through gz_gather_rows (grouped ncclSend / ncclRecv:
    every peer's block crosses its own xGMI link).  `d_local` / `d_recv` are device pointers; rows_per_rank[r] rows of
    `row_len` int32 come from rank r and land at the root in rank order."""

    def __init__(self, ctx, rank: int, world: int, uid: bytes):
        self.ctx, self.rank, self.world = ctx, rank, world
        ctx.comm_init(uid, rank, world)

    def gather_rows(self, d_local: int, rows_per_rank, row_len, root=0, d_recv: int = 0):
        self.ctx.gather_rows(d_local, int(rows_per_rank[self.rank]), row_len, d_recv, rows_per_rank, root)
        return d_recv


# ---- the fixed multi-shard job of bench.py (BASELINE configs[3]): who owns what, and the root's bookkeeping ------------
def rank_shards(rank: int, world: int, n_shards: int) -> List[int]:
    """Global shard ids rank `rank` of `world` owns: the contiguous range [n_shards * rank / world, n_shards * (rank + 1) /
    world).  `world` must divide into the shards evenly enough that every rank owns at least one."""
    if not (0 <= rank < world) or world > n_shards:
        raise ValueError("rank %d of %d for %d shards" % (rank, world, n_shards))
    return list(range(n_shards * rank // world, n_shards * (rank + 1) // world))


def global_shard_id(peer: int, local_index: int, world: int, n_shards: int) -> int:
    """Shard that arrives from rank `peer` in exchange round `local_index` (its local_index-th shard)."""
    ids = rank_shards(peer, world, n_shards)
    if not (0 <= local_index < len(ids)):
        raise ValueError("rank %d owns %d shards; round %d" % (peer, len(ids), local_index))
    return ids[local_index]


def csr_words(n_entries: int, bits: int) -> int:
    """int32 words a CSR block of `n_entries` entries of `bits` (16 | 32) bits travels as (blocks travel as whole words)."""
    if bits not in (16, 32) or n_entries < 0:
        raise ValueError("bits must be 16 or 32")
    return (n_entries * bits // 8 + 3) // 4


def block_words(n_rows: int, n_entries: int, bits: int) -> int:
    """int32 words of one rank's exchange block: [n_real[n_rows] | first[n_rows] | n_entries entries of `bits` bits]."""
    if n_rows < 0:
        raise ValueError("n_rows must not be negative")
    return 2 * n_rows + csr_words(n_entries, bits)


class GatherRound:
    """Root-side state of ONE exchange round (local shard j of every rank): where each peer's block starts in the
    receive buffer, how large the buffer must be, and what every peer announced.  A rank's block is
    [n_real[rows] | first[rows] | entries] (`block_words`): `rows[q]` is the number of rows of rank q's shard of this round --
    ranks need not own equally many."""

    def __init__(self, world: int, bits: int, rows: Sequence[int], slack: float = 1.05, pad: int = 1024):
        if len(rows) != world or any(int(r) < 0 for r in rows):
            raise ValueError("%d row counts for %d ranks" % (len(rows), world))
        self.world, self.bits, self.slack, self.pad = world, bits, slack, pad
        self.rows: List[int] = [int(r) for r in rows]
        self.capacity = 0                 # int32 words the receive buffer currently holds
        self.totals: List[int] = []       # entries announced by every rank
        self.words: List[int] = []        # int32 words of every rank's whole block (row lengths + row starts + entries)

    def worst_case_words(self, row_len: int) -> int:
        """int32 words of the round when every row of every rank is full: a receive buffer of this size never grows."""
        return sum(block_words(n, n * int(row_len), self.bits) for n in self.rows)

    def announce(self, totals: Sequence[int]) -> int:
        """Record the ENTRY counts the ranks announced for this round; returns the capacity (in int32 words) the receive
        buffer must have -- the current one when it is enough, else the grown one (the caller re-allocates when it
        differs)."""
        if len(totals) != self.world:
            raise ValueError("%d totals for %d ranks" % (len(totals), self.world))
        self.totals = [int(t) for t in totals]
        self.words = [block_words(n, t, self.bits) for n, t in zip(self.rows, self.totals)]
        need = sum(self.words)
        if need > self.capacity:
            return int(need * self.slack) + self.pad
        return self.capacity

    def word_offset(self, peer: int) -> int:
        """int32 word at which rank `peer`'s block starts in the receive buffer (blocks lie in rank order)."""
        return sum(self.words[:peer])
