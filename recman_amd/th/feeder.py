"""Pinned-memory batch feeder: the caller directly above the hot path.

The reference converts every mini-batch from pandas on the host and feeds it synchronously
(`DataInputs.load` per batch, recman/tf/inputs.py:53-58; `DeepModel.fit`, DeepModel.py:188-200).
Here a dataset is encoded ONCE; when it is kept on the host (larger than what one wants to park
in HBM) it lives in pinned memory as ONE packed record per example

    [ idx F x int64 | y 8 bytes | dense Dn x float32 | pad to 16 bytes ]

and the GPU itself pulls the rows of a batch out of it: the batch's (shuffled) row numbers go down (8 bytes
per example), rm_gather_rows - the table gather, pointed at the pinned buffer, which hipHostMalloc maps
into the GPU's address space - reads the records over PCIe in whatever order the epoch's permutation asks
for, and three slices are unpacked on the device.  Everything runs on a copy stream, `depth - 1` batches
ahead of the compute stream (HIP events both ways).  No host thread touches the rows: gathered on the
host (torch.index_select into a pinned staging slot) a 65536-row batch cost 6-13 ms, 20-40x the PCIe
time of its 17.6 MB.
"""
import torch


class BatchFeeder:
    def __init__(self, idx, dense, y, batch_size, device, depth=3):
        """idx int64 [N,F], dense float32 [N,Dn], y [N] int64 or float32 (numpy or CPU tensors)."""
        self.device = torch.device(device)
        idx = torch.as_tensor(idx).contiguous()
        dense = torch.as_tensor(dense).contiguous()
        y = torch.as_tensor(y).contiguous()
        self.n, self.F = idx.shape
        self.Dn = dense.shape[1]
        self.y_dtype = y.dtype
        self.bs = int(batch_size)
        self.depth = int(depth)
        F, Dn = self.F, self.Dn
        self.RW = -(-(2 * F + 2 + Dn) // 4) * 4  # floats per record (a multiple of 16 bytes)
        packed = torch.empty(self.n, self.RW, dtype=torch.float32).pin_memory()
        as_i64 = packed.view(torch.int64)        # [N, RW / 2]
        as_i64[:, :F] = idx
        if y.dtype == torch.int64:
            as_i64[:, F] = y
        else:
            packed[:, 2 * F] = y.to(torch.float32)
        packed[:, 2 * F + 2: 2 * F + 2 + Dn] = dense
        self.packed = packed
        from .. import ops

        self._ops = ops
        dev = self.device
        self._stage = [torch.empty(self.bs, self.RW, dtype=torch.float32, device=dev) for _ in range(self.depth)]
        self._sel = [torch.empty(self.bs, dtype=torch.int64, device=dev) for _ in range(self.depth)]
        self._dev = [(torch.empty(self.bs, F, dtype=torch.int64, device=dev),
                      torch.empty(self.bs, Dn, dtype=torch.float32, device=dev),
                      torch.empty(self.bs, dtype=y.dtype, device=dev)) for _ in range(self.depth)]
        self._copy = torch.cuda.Stream(device=dev)
        self._ready = [torch.cuda.Event() for _ in range(self.depth)]   # copy -> compute
        self._free = [torch.cuda.Event() for _ in range(self.depth)]    # compute -> copy

    def batches(self, perm=None):
        """Yields (s, t, idx_d, dense_d, y_d) for consecutive batches [s, t) of the (permuted)
        dataset; the tensors are valid until the next iteration."""
        order = (torch.arange(self.n, dtype=torch.int64) if perm is None
                 else torch.as_tensor(perm, dtype=torch.int64)).pin_memory()
        bounds = [(s, min(s + self.bs, self.n)) for s in range(0, self.n, self.bs)]
        F, Dn = self.F, self.Dn

        def issue(j):
            s, t = bounds[j]
            k, m = j % self.depth, t - s
            with torch.cuda.stream(self._copy):
                self._copy.wait_event(self._free[k])  # the compute stream is done with this slot
                sel, stage = self._sel[k][:m], self._stage[k][:m]
                sel.copy_(order[s:t], non_blocking=True)
                self._ops.gather_rows(self.packed, sel, stage)  # the GPU reads the pinned records over PCIe
                di, dd, dy = self._dev[k]
                st64 = stage.view(torch.int64)
                di[:m].copy_(st64[:, :F])
                dd[:m].copy_(stage[:, 2 * F + 2: 2 * F + 2 + Dn])
                dy[:m].copy_(st64[:, F] if self.y_dtype == torch.int64 else stage[:, 2 * F])
                self._ready[k].record(self._copy)

        cur = torch.cuda.current_stream(self.device)
        for k in range(self.depth):
            self._free[k].record(cur)
        ahead = self.depth - 1
        for j in range(min(ahead, len(bounds))):
            issue(j)
        for j, (s, t) in enumerate(bounds):
            if j + ahead < len(bounds):
                issue(j + ahead)
            k = j % self.depth
            cur.wait_event(self._ready[k])
            di, dd, dy = self._dev[k]
            yield s, t, di[: t - s], dd[: t - s], dy[: t - s]
            self._free[k].record(cur)

    def nbytes(self):
        return self.packed.numel() * 4


def encoded_nbytes(n, F, Dn):
    return n * (F * 8 + Dn * 4 + 8)
