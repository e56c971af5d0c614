"""Test-only stand-in for genz_tokenize.distributed.RcclTransport: the same gatherv contract (rows_per_rank[r] rows of
`row_len` int32 from rank r, concatenated in rank order at the root) over torch.distributed point-to-point on CPU
tensors, so that tests/test_distributed.py can run world_size 2 with the `gloo` backend.  Not part of the product
(which never imports torch)."""
import numpy as np


class GlooTransport:
    def __init__(self, rank: int, world: int):
        self.rank, self.world = rank, world

    def gather_rows(self, local: np.ndarray, rows_per_rank, row_len, root=0):
        import torch
        import torch.distributed as dist
        local = np.ascontiguousarray(local, dtype=np.int32).reshape(-1, row_len)
        assert local.shape[0] == rows_per_rank[self.rank]
        if self.rank != root:
            if local.size:
                dist.send(torch.from_numpy(local), dst=root)
            return None
        out = np.empty((int(sum(rows_per_rank)), row_len), dtype=np.int32)
        row0 = 0
        for q in range(self.world):
            k = int(rows_per_rank[q])
            if q == root:
                out[row0:row0 + k] = local
            elif k:
                buf = torch.empty((k, row_len), dtype=torch.int32)
                dist.recv(buf, src=q)
                out[row0:row0 + k] = buf.numpy()
            row0 += k
        return out
