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
