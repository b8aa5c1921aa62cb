"""Pinned-memory batch feeder: the caller directly above the hot path.

The reference converts every mini-batch from pandas on the host and feeds it synchronously
(`DataInputs.load` per batch, recman/tf/inputs.py:53-58; `DeepModel.fit`, DeepModel.py:188-200).
Here a dataset is encoded ONCE; when it is kept on the host (larger than what one wants to park
in HBM) its packed arrays live in pinned memory and the batches travel to the GPU on a copy
stream, one batch ahead of the compute stream (double-buffered device staging, HIP events both
ways).  The per-epoch shuffle is applied while a batch is gathered into its pinned staging slot.
"""
import numpy as np
import torch


class BatchFeeder:
    def __init__(self, idx, dense, y, batch_size, device, depth=2):
        """idx int64 [N,F], dense float32 [N,Dn], y [N] (numpy or CPU tensors)."""
        self.device = torch.device(device)
        self.idx = torch.as_tensor(idx).contiguous().pin_memory()
        self.dense = torch.as_tensor(dense).contiguous().pin_memory()
        self.y = torch.as_tensor(y).contiguous().pin_memory()
        self.n = self.idx.shape[0]
        self.bs = int(batch_size)
        self.depth = int(depth)
        F, Dn = self.idx.shape[1], self.dense.shape[1]
        self._host = [(torch.empty(self.bs, F, dtype=torch.int64).pin_memory(),
                       torch.empty(self.bs, Dn, dtype=torch.float32).pin_memory(),
                       torch.empty(self.bs, dtype=self.y.dtype).pin_memory()) for _ in range(self.depth)]
        self._dev = [(torch.empty(self.bs, F, dtype=torch.int64, device=self.device),
                      torch.empty(self.bs, Dn, dtype=torch.float32, device=self.device),
                      torch.empty(self.bs, dtype=self.y.dtype, device=self.device)) for _ in range(self.depth)]
        self._copy = torch.cuda.Stream(device=self.device)
        self._ready = [torch.cuda.Event() for _ in range(self.depth)]   # copy -> compute
        self._free = [torch.cuda.Event() for _ in range(self.depth)]    # compute -> copy

    def batches(self, perm=None):
        """Yields (s, t, idx_d, dense_d, y_d) for consecutive batches [s, t) of the (permuted)
        dataset; the tensors are valid until the next iteration."""
        perm_t = None if perm is None else torch.as_tensor(perm, dtype=torch.int64)
        bounds = [(s, min(s + self.bs, self.n)) for s in range(0, self.n, self.bs)]

        def issue(k, s, t):
            hi, hd, hy = self._host[k]
            if perm_t is None:
                hi[: t - s].copy_(self.idx[s:t]); hd[: t - s].copy_(self.dense[s:t]); hy[: t - s].copy_(self.y[s:t])
            else:
                sel = perm_t[s:t]
                torch.index_select(self.idx, 0, sel, out=hi[: t - s])
                torch.index_select(self.dense, 0, sel, out=hd[: t - s])
                torch.index_select(self.y, 0, sel, out=hy[: t - s])
            with torch.cuda.stream(self._copy):
                self._copy.wait_event(self._free[k])  # the compute stream is done with this slot
                di, dd, dy = self._dev[k]
                di[: t - s].copy_(hi[: t - s], non_blocking=True)
                dd[: t - s].copy_(hd[: t - s], non_blocking=True)
                dy[: t - s].copy_(hy[: t - s], non_blocking=True)
                self._ready[k].record(self._copy)

        cur = torch.cuda.current_stream(self.device)
        for k in range(self.depth):
            self._free[k].record(cur)
        for j in range(min(self.depth - 1, len(bounds))):
            issue(j % self.depth, *bounds[j])
        for j, (s, t) in enumerate(bounds):
            nxt = j + self.depth - 1
            if nxt < len(bounds):
                # the pinned staging slot is reused once its previous H2D copy has finished
                self._ready[nxt % self.depth].synchronize() if nxt >= self.depth else None
                issue(nxt % self.depth, *bounds[nxt])
            k = j % self.depth
            cur.wait_event(self._ready[k])
            di, dd, dy = self._dev[k]
            yield s, t, di[: t - s], dd[: t - s], dy[: t - s]
            self._free[k].record(cur)

    def nbytes(self):
        return self.idx.numel() * 8 + self.dense.numel() * 4 + self.y.numel() * self.y.element_size()


def encoded_nbytes(n, F, Dn):
    return n * (F * 8 + Dn * 4 + 8)
