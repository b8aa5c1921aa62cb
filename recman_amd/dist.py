"""Row-sharded embedding table across the GPUs of one node (new design: the reference
is single-process, single-device - SURVEY.md section 8e).

Layout: all per-field tables are one concatenated table; global row r lives on rank
r % W at local row r // W (cyclic, so Zipf-hot rows spread evenly).  A sharded row is
FUSED: [D embedding floats | FM bias | linear weight | pad] = D+4 floats, so one
exchange carries everything a lookup needs.  The batch stays data-parallel.

One step (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI):
  fwd  route (bucket the B*F occurrences by owner)          - rm_shard_route (counting sort)
       all_to_all of counts, then of local row ids          - 8 B per occurrence
       owner-side gather (rm_gather_rows)                   - HIP
       all_to_all of the rows back                          - (D+4)*4 B per occurrence
       FM / linear / DNN / CIN / cross exactly as on one GPU: the gather kernel reads
       occurrence (b,f) at row pos[b,f] of the received buffer (no un-route copy)
  bwd  gradient rows [dE | g | g | 0] written in bucketed order (rm_pack_grad_rows),
       all_to_all to the owners:
       each owner ends with IndexedSlices (local row ids, rows) for ITS shard;
       dense parameters: one flat all_reduce.
There is no collective in the dense compute; xGMI is a full mesh, so the all_to_all
uses all 7 links of a GPU at once.

Two exchange layouts:
  dynamic (default)   exact split sizes: a count all_to_all + ONE host sync per batch
  fixed capacity      every bucket has `cap` slots (cap = capacity_factor * n / W): equal
                      splits, no count exchange, no host sync - every shape is static, so the
                      compute between the collectives replays from hipGraphs
                      (capture_segments).  Unused slots carry id -1
                      (zero rows).  A batch that needs more than cap slots for some rank sets
                      a sticky device flag (`overflowed()`): its result must be discarded and
                      redone with the dynamic layout.  Cyclic sharding keeps uniform / hashed
                      ids within a few sigma of n/W; heavily skewed ids need the dynamic path.

Multi-valued / value features (their pooled row per example is built from several table rows): with exact split
sizes the tags travel as an expanded occurrence list behind the plain fields' occurrences (_lookup_mv); with fixed
capacity as T padded COLUMNS of the occurrence matrix (widen / _lookup_mv_fixed; id -1 = no tag, an empty occurrence
takes no slot) - static shapes, so micro-batches and captured segments work.

The routing/exchange logic below is device- and backend-agnostic torch code (it is the
same on gloo/CPU, where tests/test_dist.py runs it with world_size 2); the row gather and
routing are injected callables - HIP kernels in the product, plain torch indexing only in
the tests.
"""
import torch
import torch.distributed as dist

import os

PAD = 4  # bias, linear weight, 2 floats of padding: EXCHANGED row width D + 4 (16-byte multiple)
STATE = 4  # behind them, in the shard only: the bias / linear entries' optimizer moments (m_b m_l v_b v_l)
# shard row [D + 8 (+ alignment pad)] = [D embedding | bias | lin | m_b | m_l | v_b | v_l | pad pad] - the layout
# rm_sparse_optimizer_step_rows updates in place; only the first D + 4 floats ever travel
# RECMAN_FORCE_COLLECTIVES=1: issue the all_to_all / all_reduce calls even at world size 1
# (rehearses the RCCL path on a single GPU)
FORCE = os.environ.get("RECMAN_FORCE_COLLECTIVES", "0") == "1"


class _Done:
    """A finished collective (what the host-staged rehearsal path returns for async_op=True)."""

    def wait(self):
        return True


def _host_staged(t, group):
    """GPU tensors over the gloo backend (2+ ranks on ONE GPU: a rehearsal of the multi-rank logic
    on the real kernels, tests/test_gpu_dist.py): staged through host memory, synchronously."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_to_all(out, inp, out_splits=None, in_splits=None, group=None, async_op=False):
    if _host_staged(inp, group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(o)
        return _Done() if async_op else None
    return dist.all_to_all_single(out, inp, out_splits, in_splits, group=group, async_op=async_op)


def _all_reduce(t, group=None):
    if _host_staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
        return
    dist.all_reduce(t, group=group)


def shard_rows(R, rank, world):
    """Number of global rows r in [0, R) with r % world == rank."""
    return (R - rank + world - 1) // world


def route_torch(idx, field_off, world, cap=0):
    """Reference routing in plain torch (any device; what the CPU tests inject):
    idx [B,F], field_off [F] -> (pos [n], counts [world], send_ids [n]) with
    pos[o] = position of occurrence o in the owner-bucketed order (stable),
    send_ids[pos[o]] = local row of occurrence o on its owner.
    cap > 0: fixed-capacity layout - bucket w owns slots [w*cap, (w+1)*cap), send_ids has
    world*cap entries (-1 = empty); returns a 4th value, the overflow flag tensor."""
    g = (idx + field_off).reshape(-1)
    valid = (idx >= 0).reshape(-1)  # (an EMPTY occurrence - the padding of a tag column - takes no slot: pos -1)
    owner = torch.where(valid, g % world, torch.full_like(g, world))
    order = torch.argsort(owner, stable=True)
    counts = torch.bincount(owner, minlength=world + 1)[:world]
    pos = torch.empty_like(order)
    pos[order] = torch.arange(g.numel(), device=g.device)
    if not cap:
        nv = int(valid.sum()) if not bool(valid.all()) else g.numel()
        return torch.where(valid, pos, torch.full_like(pos, -1)), counts, (g // world)[order][:nv]
    starts = torch.cumsum(counts, 0) - counts
    own = owner.clamp(max=world - 1)
    rank_in = pos - starts[own]
    over = ((rank_in >= cap) & valid).any().to(torch.int32).reshape(1)
    ppos = own * cap + torch.clamp(rank_in, max=cap - 1)
    send = torch.full((world * cap,), -1, dtype=g.dtype, device=g.device)
    send[ppos[valid]] = (g // world)[valid]
    return torch.where(valid, ppos, torch.full_like(ppos, -1)), counts, send, over


class RowExchange:
    """The all_to_all plumbing of one batch: built once per batch from its indices, used
    for the forward row fetch and the backward gradient push."""

    def __init__(self, idx, field_off, world, route_fn, group=None, cap=0):
        self.world, self.group = world, group
        self.n = idx.numel()
        self.cap = int(cap)
        self.coll = world > 1 or (FORCE and dist.is_initialized())
        if self.cap:
            # fixed capacity: equal splits, nothing crosses to the host
            self.pos, _, self.send_ids, self.overflow = route_fn(idx, field_off, world, self.cap)
            self.slots = world * self.cap
            self.send_counts = self.recv_counts = None
            if self.coll:
                self.recv_ids = torch.empty_like(self.send_ids)
                _all_to_all(self.recv_ids, self.send_ids, group=group)
            else:
                self.recv_ids = self.send_ids
            return
        self.pos, counts, self.send_ids = route_fn(idx, field_off, world)
        self.slots = self.n
        if self.coll:
            both = torch.empty(2, world, dtype=counts.dtype, device=counts.device)
            both[0].copy_(counts)
            _all_to_all(both[1], both[0], group=group)
            # ONE host sync per batch: the split sizes of the row exchanges must be host-side
            self.send_counts, self.recv_counts = both.tolist()
            self.recv_ids = torch.empty(sum(self.recv_counts), dtype=self.send_ids.dtype,
                                        device=self.send_ids.device)
            _all_to_all(self.recv_ids, self.send_ids, self.recv_counts, self.send_counts, group=group)
        else:
            self.send_counts = self.recv_counts = [self.n]
            self.recv_ids = self.send_ids

    def fetch(self, owner_rows, async_op=False, extra_rows=0):
        """owner_rows [len(recv_ids), W]: the rows this rank serves, in recv_ids order ->
        [n, W] rows for this rank's occurrences in BUCKETED order: occurrence o is row pos[o].
        async_op: returns (rows, work) - the caller calls work.wait() before it reads the rows,
        and whatever it enqueues in between overlaps the exchange."""
        if not self.coll:
            if extra_rows:
                out = torch.empty(self.slots + extra_rows, owner_rows.shape[1], dtype=owner_rows.dtype,
                                  device=owner_rows.device)
                out[: self.slots].copy_(owner_rows)
                owner_rows = out
            return (owner_rows, None) if async_op else owner_rows
        out = torch.empty(self.slots + extra_rows, owner_rows.shape[1], dtype=owner_rows.dtype,
                          device=owner_rows.device)
        work = _all_to_all(out[: self.slots], owner_rows, self.send_counts, self.recv_counts, group=self.group,
                           async_op=async_op)
        return (out, work) if async_op else out

    def push(self, bucketed_rows, async_op=False):
        """bucketed_rows [n, W] (this rank's per-occurrence gradient rows in bucketed order)
        -> [len(recv_ids), W]: the gradient rows for the local rows recv_ids of this shard."""
        if not self.coll:
            return (bucketed_rows, None) if async_op else bucketed_rows
        out = torch.empty(len(self.recv_ids), bucketed_rows.shape[1], dtype=bucketed_rows.dtype,
                          device=bucketed_rows.device)
        work = _all_to_all(out, bucketed_rows, self.recv_counts, self.send_counts, group=self.group,
                           async_op=async_op)
        return (out, work) if async_op else out


class ShardedTable:
    """This rank's shard [R_local, D+4] of the fused table + the lookup / gradient routing."""

    def __init__(self, R, D, rank, world, device, gather_fn, route_fn, group=None, capacity_factor=None):
        self.R, self.D, self.W = R, D, D + PAD
        self.LD = D + PAD + STATE
        if self.LD % 32 and 32 - self.LD % 32 <= 8:
            # D = 16: 24 -> 32 floats, one 128-byte line per shard row.  At 24 every second row straddles two
            # lines and the owner-side gather (and the optimizer's read-modify-write) paid 1.5 line requests
            # per row: rm_gather_rows 90 -> 66 us for 1.7 M rows
            self.LD += 32 - self.LD % 32
        self.rank, self.world, self.group = rank, world, group
        self.capacity_factor = capacity_factor  # None: dynamic split sizes
        self.shard = torch.zeros(shard_rows(R, rank, world), self.LD, dtype=torch.float32, device=device)
        self.gather_fn, self.route_fn = gather_fn, route_fn

    def init_reference(self, offsets, sizes, seed=2019):
        """Initial values with the reference's distribution for THIS rank's rows: row r of feature f ~
        truncated normal(0, sqrt(2 / (V_f + D))) (FeatEmbedding._upsert_variables, layers.py:95-110;
        glorot_normal utils.py:180-183), bias and linear entries zero.  Drawn per rank (seed + rank): TF's
        stream cannot be reproduced anyway, and the table never exists in one piece."""
        dev = self.shard.device
        g = torch.Generator(device=dev).manual_seed(int(seed) + 7919 * self.rank)
        n = self.shard.shape[0]
        glob = torch.arange(n, device=dev, dtype=torch.int64) * self.world + self.rank
        offs = torch.tensor(list(offsets), device=dev, dtype=torch.int64)
        feat = torch.searchsorted(offs, glob, right=True) - 1
        std = torch.sqrt(2.0 / (torch.tensor(list(sizes), device=dev, dtype=torch.float32)[feat] + self.D))
        self.shard.zero_()
        for s0 in range(0, n, 1 << 22):  # in chunks: the shard is 12.5 M rows at BASELINE configs[4]
            s1 = min(n, s0 + (1 << 22))
            z = torch.empty(s1 - s0, self.D, device=dev)
            torch.nn.init.trunc_normal_(z, 0.0, 1.0, -2.0, 2.0, generator=g)
            self.shard[s0:s1, : self.D] = z * std[s0:s1, None]

    def load_global(self, table, bias=None, lin=None):
        """Fills the shard from full-size arrays (tests / small tables): row r -> rank r % W."""
        sl = slice(self.rank, self.R, self.world)
        self.shard[:, : self.D] = table[sl].to(self.shard.device)
        if bias is not None:
            self.shard[:, self.D] = bias[sl].to(self.shard.device)
        if lin is not None:
            self.shard[:, self.D + 1] = lin[sl].to(self.shard.device)

    # ---- checkpoint: one file per rank (the shard never has to fit one host buffer) ----
    @staticmethod
    def shard_path(path, rank, world):
        return f"{path}.shard{rank}of{world}.pt"

    def save(self, path):
        """Writes this rank's shard to `<path>.shard<rank>of<world>.pt` (every rank calls it): the PARAMETER
        columns [D embedding | bias | linear] only - the row padding and the optimizer-state columns behind them
        are layout details of this build (their width changed between rounds) and optimizer state is not part of
        a model checkpoint (the embedding moments live in another array anyway)."""
        torch.save({"R": self.R, "D": self.D, "rank": self.rank, "world": self.world, "width": self.D + 2,
                    "rows": self.shard[:, : self.D + 2].cpu().contiguous()},
                   self.shard_path(path, self.rank, self.world))

    def _check(self, ck):
        if (ck["R"], ck["D"]) != (self.R, self.D):
            raise ValueError(f"checkpoint table is {ck['R']}x{ck['D']}, this one {self.R}x{self.D}")
        w = ck["rows"].shape[1]
        if w < self.D + 2:
            raise ValueError(f"checkpoint rows have {w} columns, expected at least D + 2 = {self.D + 2}")
        return ck["rows"][:, : self.D + 2]  # (older files carry padding / state columns behind the parameters)

    def load(self, path, saved_world=None):
        """Restores the shard's parameters; every optimizer-state column of the rows is reset to zero (a restored
        model starts its optimizer afresh: the dense moments and the step count do too).  Same world size: reads
        this rank's own file.  Different world size (re-sharding): streams the saved shards one at a time and
        keeps the rows that now belong here (global row r = local * saved_world + saved_rank; it lives here when
        r % world == rank)."""
        w0 = saved_world or self.world
        P = self.D + 2
        self.shard[:, P:].zero_()
        if w0 == self.world:
            ck = torch.load(self.shard_path(path, self.rank, self.world), weights_only=True)
            self.shard[:, :P].copy_(self._check(ck))
            if getattr(self, "on_load", None) is not None:
                self.on_load()
            return
        for r0 in range(w0):
            ck = torch.load(self.shard_path(path, r0, w0), weights_only=True)
            rows = self._check(ck)
            g = torch.arange(rows.shape[0], dtype=torch.int64) * w0 + r0  # global row ids
            mine = (g % self.world) == self.rank
            self.shard[(g[mine] // self.world).to(self.shard.device), :P] = rows[mine].to(self.shard.device)
        if getattr(self, "on_load", None) is not None:
            self.on_load()  # (the optimizer re-initialises the state columns it keeps in the rows)

    def lookup(self, idx, field_off, extra_rows=0):
        """idx [B,F] -> (rows [n, D+4] in BUCKETED order, the RowExchange): the row of
        occurrence o = b*F+f is rows[ex.pos[o]] - consumers gather through pos, no un-route copy.
        extra_rows: that many more rows are allocated behind the received ones (the pooled rows of
        multi-valued features are built there, addressable like received rows)."""
        ex = self.lookup_start(idx, field_off, extra_rows)
        return self.lookup_finish(ex), ex

    def lookup_start(self, idx, field_off, extra_rows=0):
        """Routes, exchanges the ids, gathers this shard's rows and STARTS the row exchange; what
        the caller enqueues before lookup_finish(ex) runs while the rows travel."""
        ex = RowExchange(idx, field_off, self.world, self.route_fn, self.group, self.capacity(idx.numel()))
        served = torch.empty(len(ex.recv_ids), self.W, dtype=torch.float32, device=self.shard.device)
        self.gather_fn(self.shard[:, : self.W], ex.recv_ids, served)  # (the state columns stay home)
        ex.served = served  # keeps the send buffer alive until the exchange is done
        ex.rows, ex.rows_work = ex.fetch(served, async_op=True, extra_rows=extra_rows)
        return ex

    def lookup_finish(self, ex):
        if ex.rows_work is not None:
            ex.rows_work.wait()  # the current stream waits; the host does not
            ex.rows_work = None
        return ex.rows

    def capacity(self, n):
        """Slots per owner bucket for n occurrences (0 = dynamic layout): capacity_factor * n / W
        plus 6 sigma of the binomial spread, rounded up to 64."""
        if not self.capacity_factor:
            return 0
        n = getattr(self, "cap_occurrences", None) or n  # (fit(): the largest part over the ranks - rank-uniform)
        mean = n / self.world
        cap = self.capacity_factor * mean + 6.0 * (mean * (1 - 1 / self.world)) ** 0.5 + 1
        return int(-(-cap // 64) * 64)

    def push_grads(self, ex, bucketed_grads):
        """bucketed_grads [n, D+4] (row pos[o] = gradient of occurrence o) -> (local row ids,
        gradient rows) for this shard: IndexedSlices, duplicates not merged."""
        return ex.recv_ids, ex.push(bucketed_grads)


_DENSE_L2 = ("deep_l2_reg", "cin_l2_reg", "cross_layer_l2_reg")


def flatten_grads(grads):
    """Re-homes every dense-parameter gradient as a view of ONE flat buffer (16-byte aligned
    slots) so the data-parallel all_reduce runs in place, without a gather/scatter copy per
    parameter.  Mutates `grads`; returns the flat buffer."""
    keys = sorted(grads)
    offs, n = [], 0
    for k in keys:
        offs.append(n)
        n += (grads[k].numel() + 3) // 4 * 4
    any_g = grads[keys[0]]
    flat = torch.zeros(n, dtype=any_g.dtype, device=any_g.device)
    for k, o in zip(keys, offs):
        v = flat[o: o + grads[k].numel()].view_as(grads[k])
        v.copy_(grads[k])
        grads[k] = v
    return flat


def allreduce_dense(grads, world, group=None, flat=None, average=True):
    """One flat all_reduce over every dense-parameter gradient (sum / world, or the plain sum
    when the per-rank gradients already carry the 1/world factor); in place when the gradients
    already live in `flat` (flatten_grads)."""
    if world == 1 and not (FORCE and dist.is_initialized()):
        return
    if flat is not None:
        _all_reduce(flat, group)
        if world > 1 and average:
            flat.div_(world)
        return
    keys = sorted(grads)
    flat = torch.cat([grads[k].reshape(-1) for k in keys])
    _all_reduce(flat, group)
    if average:
        flat.div_(world)
    off = 0
    for k in keys:
        n = grads[k].numel()
        grads[k].copy_(flat[off: off + n].view_as(grads[k]))
        off += n


def hip_gather(table, rows, out):
    from . import ops

    ops.gather_rows(table, rows, out)


class HipRouter:
    """rm_shard_route with its buffers (counting sort by owner on the GPU).  The outputs rotate
    through `ring` buffer sets: with micro-batching the routing of batch c+1 is computed while
    batch c still reads its own pos / ids."""

    def __init__(self, device, ring=1):
        self.device = device
        self.ring = max(1, int(ring))
        self._n = None
        self._turn = 0

    def ensure(self, n, world, cap=0):
        """Allocates the output ring, the sticky overflow flag and the workspace for n occurrences."""
        from . import ops

        if self._n != (n, world, cap):
            self._n = (n, world, cap)
            dev = self.device
            self._sets = [(torch.empty(n, dtype=torch.int64, device=dev),
                           torch.empty(world * cap if cap else n, dtype=torch.int64, device=dev),
                           torch.empty(world, dtype=torch.int64, device=dev)) for _ in range(self.ring)]
            self.overflow = torch.zeros(1, dtype=torch.int32, device=dev)  # sticky
            self.ws = torch.empty(ops._lib.lib().rm_shard_route_workspace(world), dtype=torch.int32,
                                  device=dev)

    def __call__(self, idx, field_off, world, cap=0):
        from . import ops

        self.ensure(idx.numel(), world, cap)
        self._turn = (self._turn + 1) % self.ring
        self.pos, self.ids, self.counts = self._sets[self._turn]
        if cap:
            ops.shard_route_padded(idx, field_off, world, cap, self.pos, self.ids, self.counts,
                                   self.overflow, self.ws)
            return self.pos, self.counts, self.ids, self.overflow
        ops.shard_route(idx, field_off, world, self.pos, self.ids, self.counts, self.ws)
        return self.pos, self.counts, self.ids


def make_sharded_engine(model, spec, D, hp, device, rank, world, group=None, capacity_factor=None,
                        micro_batches=1, mv_capacity=None):
    """An engine whose embedding table is row-sharded over `world` ranks (bench.py --gpus N).
    capacity_factor: fixed-capacity exchange layout, see the module doc.  micro_batches > 1:
    the step is software-pipelined over that many micro-batches so that the row / gradient-row
    exchanges overlap the dense compute.  mv_capacity: name -> T, the most tags an example carries in a
    multi-valued feature (the SAME numbers on every rank; value features: 1): with the fixed-capacity layout
    the tags travel as T padded columns of the occurrence matrix (set_mv_capacity)."""
    from . import engine as eng

    base = eng.ENGINES[model]

    class Sharded(base):
        sharded = True
        step_fusable = False  # (DeepFM's one-kernel step reads a LOCAL table; here the rows arrive by exchange)

        def __init__(self):
            if spec.scratch_names and int(micro_batches) > 1 and not capacity_factor:
                # with exact split sizes their tags travel as an expanded occurrence list whose length changes
                # from batch to batch: no micro-batch pipeline there (the fixed-capacity layout has one)
                raise NotImplementedError(
                    f"row-sharded table: multi-valued / value features {sorted(spec.scratch_names)} with "
                    "micro_batches > 1 need the fixed-capacity exchange layout (capacity_factor)")
            self._mv_T = dict(mv_capacity or {})
            self._wide = None
            for k in ("embedding_l2_reg", "linear_l2_reg"):
                if hp.get(k, 0.0) and not hp.get("lazy_l2", True):
                    # the reference's DENSE l2 term on the table makes EVERY row's gradient non-zero
                    # (layers.py:188-193): 25.6 GB per step at BASELINE configs[4] - the row-wise owner-side step
                    # cannot honour it; the lazy form (reg * row for the rows a step touches, applied by
                    # ShardedOptimizer) is what this engine offers
                    raise NotImplementedError(f"row-sharded table: {k} needs lazy_l2 (a dense table gradient otherwise)")
            self._shard_args = (rank, world, group)
            self._pending = None
            self._slot = None
            self._segs = None
            self.micro_batches = int(micro_batches)
            super().__init__(spec, D, hp, device=device)
            self._hp_full = dict(self.hp)
            self._flat_grads = flatten_grads(self.grads)
            self._acc = torch.zeros_like(self._flat_grads)

        def _alloc_tables(self):
            dev = self.device
            R = self.spec.rows
            self.st = ShardedTable(R, self.D, rank, world, dev, hip_gather,
                                   HipRouter(dev, ring=self.micro_batches + 1), group, capacity_factor)
            self.table = self.st.shard  # [R_local, D+8]: fused rows [D | bias | lin | 2 pad] + 4 state columns
            self._set_lin_masks()       # linear_features subsets: masked per field in the gradient rows
            self.linear_w_dense = torch.zeros(self.Dn, dtype=torch.float32, device=dev)
            self.field_off = torch.tensor(self.spec.offsets(), dtype=torch.int64, device=dev)
            self.lin_off = self.field_off
            self.params["table_shard"] = self.st.shard
            self.params["linear_w_dense"] = self.linear_w_dense

        def _alloc_mv(self, B):
            # (no scratch block beside a local table: the pooled rows are built behind the received rows)
            self._arange = torch.arange(B, dtype=torch.int64, device=self.device)
            self._fix_cols = [f for f in range(self.F) if f not in self.mv_fields]
            self._mv_layout()

        # ---- multi-valued / value features under the fixed-capacity layout: tags as padded COLUMNS ----
        def set_mv_capacity(self, caps):
            """name -> T (most tags per example; the same on every rank).  Buffers are re-made on the next batch."""
            caps = dict(caps)
            if caps != self._mv_T:
                self._mv_T, self._B, self._segs = caps, None, None

        def _mv_fixed(self):
            return bool(getattr(self, "mv_fields", None)) and bool(self.st.capacity_factor)

        def exchange_columns(self):
            """Columns of the occurrence matrix one example sends through the exchange: F, or F_wide (the plain
            fields + the padded tag columns) for scratch-row features under the fixed-capacity layout."""
            names = self.spec.sparse_names
            scratch = [n for n in names if n in self.spec.scratch_names]
            if not (scratch and self.st.capacity_factor):
                return self.F
            return (self.F - len(scratch)
                    + sum(1 if n in self.spec.value_names else int(self._mv_T.get(n, 0)) for n in scratch))

        def _mv_layout(self):
            """The WIDE occurrence matrix [B, F_wide]: the plain fields' columns, then T_f columns per scratch-row
            feature f holding its tags (-1 = none).  Every column has a field offset, so routing, the exchange,
            the owner side and the optimizer treat a tag like any other occurrence; shapes are static - micro-
            batches are row slices and the step can be captured."""
            names = self.spec.sparse_names
            self._mv_cols, c = [], len(self._fix_cols)
            foff = [int(self.field_off[f]) for f in self._fix_cols]
            for f in self.mv_fields:
                T = 1 if names[f] in self.spec.value_names else int(self._mv_T.get(names[f], 0))
                self._mv_cols.append((c, T))
                foff += [int(self.field_off[f])] * T
                c += T
            self.F_wide = c
            self._lin_mask_host = self.spec.lin_masks()[0]  # (host copy: no device read inside captured segments)
            self._foff_wide = torch.tensor(foff, dtype=torch.int64, device=self.device)
            self._fix_t = torch.tensor(self._fix_cols, dtype=torch.int64, device=self.device)

        def widen(self, idx, mv, out=None):
            """(idx [B, F], mv dict of CSR entries) -> (idx_wide [B, F_wide] int64, vals_wide [B, F_wide] float32):
            tag t of example b of feature f in column c0_f + t.  torch ops over the CSR (its length varies): this
            runs OUTSIDE the captured segments, which read the static wide buffers."""
            B = idx.shape[0]
            if out is None:
                out = (torch.empty(B, self.F_wide, dtype=torch.int64, device=self.device),
                       torch.zeros(B, self.F_wide, dtype=torch.float32, device=self.device))
            iw, vw = out
            nfix = len(self._fix_cols)
            iw[:, :nfix] = idx.index_select(1, self._fix_t)
            iw[:, nfix:] = -1
            vw[:, nfix:] = 0
            self._mv = mv
            rows = torch.arange(B, dtype=torch.int64, device=self.device)
            for (c0, T), f in zip(self._mv_cols, self.mv_fields):
                offsets, ids, vals = self._mv_entry(f)
                n = offsets[1:] - offsets[:-1]
                if ids.numel() and int(n.max()) > T:
                    raise ValueError(f"feature {self.spec.sparse_names[f]}: an example carries {int(n.max())} tags, "
                                     f"mv_capacity allows {T}")
                seg = torch.repeat_interleave(rows, n)
                col = c0 + torch.arange(ids.numel(), dtype=torch.int64, device=self.device) - offsets[seg]
                iw[seg, col] = ids
                if vals is not None:
                    vw[seg, col] = vals
            return iw, vw

        def _alloc(self, B):
            first = self._B != B
            super()._alloc(B)
            if first:
                self._zoff = torch.zeros(self.F, dtype=torch.int64, device=self.device)
                self._zoff1 = torch.zeros(1, dtype=torch.int64, device=self.device)
                Fx = self.F_wide if self._mv_fixed() else self.F
                slots = world * self.st.capacity(B * Fx) or B * Fx
                extra = len(self.mv_fields) * B if self._mv_fixed() else 0  # (the pooled rows' own gradients stay home)
                self.grad_rows = torch.zeros(slots + extra, self.D + PAD, dtype=torch.float32, device=self.device)
                self._pos_model = torch.empty(B, self.F, dtype=torch.int64, device=self.device)
                # one send buffer per micro-batch: its exchange is still in flight while the
                # next micro-batch packs
                self.grad_rows_m = [self.grad_rows] + [torch.empty_like(self.grad_rows)
                                                       for _ in range(self.micro_batches - 1)]

        def _front_fused(self, m, lin_w):
            # the one-kernel front gathers from the RECEIVED rows (stride D + 4) through their positions;
            # scratch-row features keep the two-kernel path (their pooled rows are built in between)
            return not self.mv_fields and base._front_fused(self, m, lin_w)

        def _front_ld(self):
            return self.D + PAD

        def _front_table(self, idx):
            pos = self._lookup(idx)
            return pos, self.rows, self._zoff, self.D + PAD, False

        def _lookup(self, idx):
            """Runs / finishes this (micro-)batch's row exchange; sets self.ex / self.rows and returns pos [B, F]:
            occurrence (b, f) is row pos[b, f] of self.rows."""
            B = idx.shape[0]
            if self.mv_fields:
                pos = self._lookup_mv_fixed(idx) if (self._mv_fixed() and self._wide is not None) else self._lookup_mv(idx)
            elif self._slot is not None:      # captured segment: static buffers, exchange done by the caller
                self.ex, self.rows = self._slot, self._slot.rows
                pos = self.ex.pos.view(B, self.F)
            elif self._pending is not None:   # a micro-batch whose exchange was started earlier
                self.ex, self._pending = self._pending, None
                self.rows = self.st.lookup_finish(self.ex)
                pos = self.ex.pos.view(B, self.F)
            else:
                self.rows, self.ex = self.st.lookup(idx, self.field_off)
                pos = self.ex.pos.view(B, self.F)
            self._pos = pos
            return pos

        def _embed(self, idx, dense, want_fm, masks, lin_w=None):
            from . import ops

            m = masks or {}
            fm_masks = m.get("fm", (None, None))
            B = idx.shape[0]
            # rows arrive owner-bucketed; the gather kernel reads occurrence (b,f) at row pos[b,f]
            pos = self._lookup(idx)
            W = self.D + PAD
            flat = self.rows.view(-1)
            ops.embed_fwd(
                pos, self.rows, self._zoff, table_ld=W, D=self.D,
                bias_table=flat[self.D:] if want_fm else None, bias_ld=W,
                lin_w=flat[self.D + 1:] if self.use_linear else None, lin_ld=W, lin_off=self._zoff,
                lin_w_dense=self.linear_w_dense if (self.use_linear and self.Dn) else None,
                lin_w0=self.params["linear_w0"] if self.use_linear else None,
                dense=dense if (self.use_linear and self.Dn) else None,
                mask_b=fm_masks[0] if want_fm else None, mask_e=fm_masks[1] if want_fm else None,
                E=self.E, fm_sum=self.fm_sum if want_fm else None,
                fm_logit=self.fm_logit if want_fm else None,
                lin_logit=self.lin_logit if self.use_linear else None)

        def _lookup_mv(self, idx):
            """Multi-valued / value features: the exchange carries ONE flat occurrence list - the plain
            fields' (b, f) occurrences followed by every tag of every scratch-row feature - and each
            example's pooled row (rm_pool_rows: sqrtn combiner, or value-weighted) is built from the
            received tag rows into rows BEHIND the received ones; returns pos [B, F] for the gather."""
            from . import ops

            B, D, W = idx.shape[0], self.D, self.D + PAD
            fix = self._fix_cols
            lists = [(idx[:, fix] + self.field_off[fix]).reshape(-1)]
            ents = []
            for f in self.mv_fields:
                offsets, ids, vals = self._mv_entry(f)
                ents.append((offsets, ids, vals))
                lists.append(ids + self.field_off[f])
            flat = torch.cat(lists).view(-1, 1)
            nmv = len(self.mv_fields)
            cf, self.st.capacity_factor = self.st.capacity_factor, None  # (a list of varying length: exact splits)
            try:
                self.rows, self.ex = self.st.lookup(flat, self._zoff1, extra_rows=nmv * B)
            finally:
                self.st.capacity_factor = cf
            slots = self.ex.slots
            pos = torch.empty(B, self.F, dtype=torch.int64, device=idx.device)
            nfix = B * len(fix)
            pos[:, fix] = self.ex.pos[:nfix].view(B, len(fix))
            self._mv_pos, o = [], nfix
            for j, (f, (offsets, ids, vals)) in enumerate(zip(self.mv_fields, ents)):
                ptag = self.ex.pos[o: o + ids.numel()]
                o += ids.numel()
                self._mv_pos.append(ptag)
                tag_rows = self.rows.index_select(0, ptag)
                if vals is None:
                    tag_rows[:, D + 1] *= (ids >= 1).to(tag_rows.dtype)  # slot 0 leaves the multi-hot count (utils.py:108)
                # (ids shifted by one against row0 = -1: rm_pool_rows' own `id >= 1` rule then keeps every row)
                ops.pool_rows(tag_rows, -1, D, offsets, torch.arange(1, ids.numel() + 1, device=ids.device), 
                              self.rows[slots + j * B: slots + (j + 1) * B], vals=vals)
                pos[:, f] = slots + j * B + self._arange
            return pos

        def _lookup_mv_fixed(self, idx):
            """The fixed-capacity form of _lookup_mv: the exchange carried the WIDE occurrence matrix (self._wide);
            the pooled rows are built by rm_pool_rows_padded behind the received rows."""
            from . import ops

            B, D = idx.shape[0], self.D
            nmv = len(self.mv_fields)
            iw, vw = self._wide
            if self._slot is not None:        # captured segment: static buffers, exchange done by the caller
                self.ex, self.rows = self._slot, self._slot.rows
            elif self._pending is not None:   # a micro-batch whose exchange was started earlier
                self.ex, self._pending = self._pending, None
                self.rows = self.st.lookup_finish(self.ex)
            else:
                self.rows, self.ex = self.st.lookup(iw, self._foff_wide, extra_rows=nmv * B)
            slots = self.ex.slots
            posw = self.ex.pos.view(B, self.F_wide)
            pos = self._pos_model
            pos.index_copy_(1, self._fix_t, posw[:, : len(self._fix_cols)])
            names = self.spec.sparse_names
            for j, ((c0, T), f) in enumerate(zip(self._mv_cols, self.mv_fields)):
                ops.pool_rows_padded(self.rows, D, posw[:, c0: c0 + T], iw[:, c0: c0 + T],
                                     self.rows[slots + j * B: slots + (j + 1) * B],
                                     vals=vw[:, c0: c0 + T] if names[f] in self.spec.value_names else None)
                pos[:, f] = self._arange + (slots + j * B)
            self._posw = posw
            return pos

        def _pack_mv_fixed(self, grad_rows):
            """Gradient rows of one (micro-)batch into the send buffer: the plain fields' (and, behind the slots,
            the pooled rows' own) by rm_pack_grad_rows, every tag's by rm_pack_pooled_grad_rows."""
            from . import ops

            g_fm = self.dlogit if self._has_fm() else None
            ops.pack_grad_rows(self.d_rows, g_fm, self.dlogit if self.use_linear else None,
                               self._pos_model.reshape(-1), grad_rows, lin_field_mask=self.lin_field_mask)
            iw, vw = self._wide
            names = self.spec.sparse_names
            for (c0, T), f in zip(self._mv_cols, self.mv_fields):
                lin_on = self.use_linear and (self._lin_mask_host is None or float(self._lin_mask_host[f]) != 0.0)
                ops.pack_pooled_grad_rows(self.d_rows[:, f, :], g_fm, self.dlogit if lin_on else None, self.D,
                                          self._posw[:, c0: c0 + T], iw[:, c0: c0 + T], grad_rows,
                                          vals=vw[:, c0: c0 + T] if names[f] in self.spec.value_names else None)

        def _pack_mv(self, grad_rows):
            """Gradient rows of the tags: each tag occurrence has its own slot; its row is the pooled row's
            gradient times the tag's pooling factors (what rm_pool_rows_bwd scatters on one GPU)."""
            B, D = self._B, self.D
            g_fm = self.dlogit if self._has_fm() else None
            for j, f in enumerate(self.mv_fields):
                offsets, ids, vals = self._mv_entry(f)
                n = offsets[1:] - offsets[:-1]
                seg = torch.repeat_interleave(self._arange, n)
                if vals is not None:   # value feature: embedding and linear scaled by the value, bias not
                    we, wb, wl = vals, torch.ones_like(vals), vals
                else:                  # sqrtn combiner; the linear term is a multi-hot count without slot 0
                    inv = n.clamp(min=1).to(torch.float32).rsqrt()[seg]
                    we, wb, wl = inv, inv, (ids >= 1).to(torch.float32)
                g = torch.zeros(ids.numel(), D + PAD, dtype=torch.float32, device=ids.device)
                g[:, :D] = self.d_rows[seg, f, :] * we.unsqueeze(1)
                if g_fm is not None:
                    g[:, D] = g_fm[seg] * wb
                lin_on = self.lin_field_mask is None or float(self.lin_field_mask[f]) != 0.0
                if self.use_linear and lin_on:
                    g[:, D + 1] = self.dlogit[seg] * wl
                grad_rows.index_copy_(0, self._mv_pos[j], g)

        def _step_packed_ok(self, masks, mv):
            """DeepFM's one-kernel step (rm_deepfm_step, packed form) covers this (micro-)batch: it gathers from the
            RECEIVED rows and writes the gradient rows straight into the send buffer of the backward exchange -
            the front kernel, the backward kernel and rm_pack_grad_rows in one launch."""
            from . import ops

            if self.model != "deepfm" or masks or mv is not None or self.mv_fields:
                return False
            if not self.hp.get("step_fusion", eng.STEP_FUSION_DEFAULT):
                return False
            if not (self.use_fm and self.use_deep and self.use_linear and self.use_bias_tables):
                return False
            if getattr(self, "_step_pk", None) is None:
                self._step_pk = bool(self.mlp.fused_ok and ops.deepfm_step_supported(
                    self.F, self.D, self.D + PAD, self.Dn, self.mlp.hidden))
            return self._step_pk

        def _fwd_bwd_packed(self, idx, dense, y, masks, grad_rows, mv=None, pos=None):
            """Forward + backward of one (micro-)batch with its gradient rows [dE | g_fm | g_lin | 0 0] written in
            bucketed order into grad_rows (the send buffer of the backward all_to_all); returns the loss."""
            from . import ops

            if self._step_packed_ok(masks, mv):
                B = idx.shape[0]
                self._alloc(B)
                self._mv = None
                pos = self._lookup(idx).view(B, self.F)  # runs / finishes the row exchange: self.rows
                mlp, p, g = self.mlp, self.params, self.grads
                if getattr(self, "_step_ws", None) is None:
                    self._step_ws = torch.zeros(ops.deepfm_step_workspace(self.F, self.Dn), dtype=torch.float32,
                                                device=self.device)
                pre, n = mlp.prefix, len(mlp.hidden)
                rows = self.rows
                ops.deepfm_step(
                    pos, rows, self._zoff, self.D, self.D + PAD, dense if self.Dn else None, y,
                    [p[f"{pre}dnn_layer_{i}_weights"] for i in range(n)],
                    [p[f"{pre}dnn_layer_{i}_bias"] for i in range(n)], p[f"{pre}dnn_w"].view(-1), p[f"{pre}dnn_w0"],
                    self.linear_w_dense if self.Dn else None, p["linear_w0"], mlp.act, self.task, grad_rows,
                    self.logit, self.pred, self.dlogit, self.loss,
                    [g[f"{pre}dnn_layer_{i}_weights"] for i in range(n)],
                    [g[f"{pre}dnn_layer_{i}_bias"] for i in range(n)], g[f"{pre}dnn_w"].view(-1), g[f"{pre}dnn_w0"],
                    g["linear_w_dense"] if self.Dn else None, g["linear_w0"], self._step_ws,
                    grad_scale=getattr(self, "grad_scale", 1.0), packed_rows=min(rows.shape[0], grad_rows.shape[0]),
                    lin_field_mask=self.lin_field_mask)
                self.d_bias = None
                if self.Dn and self.lin_dense_mask is not None:
                    g["linear_w_dense"].mul_(self.lin_dense_mask)
                reg = self.hp.get("deep_l2_reg", 0.0)
                if reg:
                    mlp.add_l2_grads(reg)
                return self._add_l2(self.loss)
            loss = base.fwd_bwd(self, idx, dense, y, masks, mv=mv) if mv is not None else base.fwd_bwd(self, idx, dense, y, masks)
            if self.mv_fields:
                if self._mv_fixed() and self._wide is not None:
                    self._pack_mv_fixed(grad_rows)
                return loss
            ops.pack_grad_rows(self.d_rows, self.dlogit if self._has_fm() else None,
                               self.dlogit if self.use_linear else None,
                               (self._pos if pos is None else pos).reshape(-1), grad_rows,
                               lin_field_mask=self.lin_field_mask)
            return loss

        def _one(self, idx, dense, y, masks, grad_rows, mv=None):
            """fwd+bwd of one (micro-)batch; its gradient rows are packed into grad_rows and their
            exchange is started: returns (loss, ids, rows, work)."""
            from . import ops

            loss = self._fwd_bwd_packed(idx, dense, y, masks, grad_rows, mv=mv)
            if self.mv_fields and self._mv_fixed() and self._wide is not None:
                out, work = self.ex.push(grad_rows[: self.ex.slots], async_op=True)
                return loss, self.ex.recv_ids, out, work
            if self.mv_fields:
                # (slots + pooled-row region; the pooled rows' own gradients land behind the slots and stay home)
                grad_rows = torch.zeros(self.rows.shape[0], self.D + PAD, dtype=torch.float32, device=self.device)
                # gradient rows [dE | g_fm | g_lin | 0 0], written straight in bucketed order -> owners
                ops.pack_grad_rows(self.d_rows, self.dlogit if self._has_fm() else None,
                                   self.dlogit if self.use_linear else None, self._pos.reshape(-1), grad_rows,
                                   lin_field_mask=self.lin_field_mask)
            if self.mv_fields:
                self._pack_mv(grad_rows)
                grad_rows = grad_rows[: self.ex.slots]
            out, work = self.ex.push(grad_rows, async_op=True)
            return loss, self.ex.recv_ids, out, work

        def fwd_bwd(self, idx, dense, y, masks=None, weight=None, mv=None):
            """One step.  weight: this rank's share of the global batch (default 1 / world: equal
            per-rank batches; fit() passes B_local / B_global for a ragged last batch).
            micro_batches = M > 1 splits the batch into M equal micro-batches and
            software-pipelines them: while micro-batch c computes, the rows of c+1 and the gradient
            rows of c-1 travel over xGMI (RCCL runs them on its own stream).  The gradients are the
            full-batch means either way: shard_grad_ids / shard_grad_rows (lists of M IndexedSlices
            pieces when M > 1) and self.grads."""
            M = self.micro_batches
            B = idx.shape[0]
            if self._segs is not None and masks is None and weight is None:
                return self._replay_segments(idx, dense, y, mv)
            self._alloc(B if M <= 1 else B // M)
            # scratch-row features under the fixed-capacity layout: their tags as padded columns (widen)
            wide = self.widen(idx, mv) if (self._mv_fixed() and mv is not None) else None
            # every gradient is the gradient of the GLOBAL batch mean: each rank's (micro-)batch
            # carries its share `weight` / M (1 / world for equal per-rank batches), the owners sum the
            # rows they receive, the dense all_reduce sums.  The l2 terms of the dense parameters are
            # added ONCE: each (micro-)batch adds reg * weight / M of them (hp scaled for the call), and the
            # shares add up to reg over ranks and micro-batches.
            w_rank = (1.0 / world) if weight is None else float(weight)
            self.grad_scale = w_rank / M
            self.hp = dict(self._hp_full, **{k: self._hp_full.get(k, 0.0) * self.grad_scale for k in _DENSE_L2})
            if M <= 1:
                try:
                    self._wide = wide
                    loss, ids, rows, work = self._one(idx, dense, y, masks, self.grad_rows, mv=mv)
                finally:
                    self.hp = self._hp_full
                    self._wide = None
                if work is not None:
                    work.wait()
                self.shard_grad_ids, self.shard_grad_rows = ids, rows
                allreduce_dense(self.grads, world, group, self._flat_grads, average=False)
                return loss + self._l2_once()
            if B % M or masks is not None:
                raise ValueError("micro-batching needs a batch divisible by micro_batches and no dropout masks")
            b = B // M
            parts = [(idx[c * b: (c + 1) * b], dense[c * b: (c + 1) * b], y[c * b: (c + 1) * b]) for c in range(M)]
            if self.mv_fields and wide is None:
                raise ValueError("micro-batching with multi-valued / value features needs their mv= entries")
            wides = [None] * M if wide is None else [(wide[0][c * b: (c + 1) * b], wide[1][c * b: (c + 1) * b])
                                                     for c in range(M)]

            def start(c):
                if wides[c] is None:
                    return self.st.lookup_start(parts[c][0], self.field_off)
                return self.st.lookup_start(wides[c][0], self._foff_wide, extra_rows=len(self.mv_fields) * b)

            started = start(0)
            outs, total = [], None
            for c in range(M):
                self._pending = started
                # start the NEXT micro-batch's exchange before this one's compute is enqueued
                started = start(c + 1) if c + 1 < M else None
                try:
                    self._wide = wides[c]
                    loss, ids, rows, work = self._one(*parts[c], None, self.grad_rows_m[c])
                except BaseException:
                    self.hp = self._hp_full
                    raise
                finally:
                    self._wide = None
                outs.append((ids, rows, work))
                if c == 0:
                    self._acc.copy_(self._flat_grads)
                    total = loss.clone()
                else:
                    self._acc.add_(self._flat_grads)
                    total.add_(loss)
            self._flat_grads.copy_(self._acc)  # every chunk already carries the 1/(world*M) factor
            for _, _, work in outs:
                if work is not None:
                    work.wait()
            self.hp = self._hp_full
            self.shard_grad_ids = [o[0] for o in outs]
            self.shard_grad_rows = [o[1] for o in outs]
            allreduce_dense(self.grads, world, group, self._flat_grads, average=False)
            return total.div_(M) + self._l2_once()

        def _add_l2(self, loss):
            return loss  # (the data loss only: the dense parameters' l2 value is added once per step)

        def _l2_once(self):
            """The l2 VALUE of the dense parameters with the full coefficients (identical on every rank)."""
            if not any(self._hp_full.get(k, 0.0) for k in _DENSE_L2):
                return 0.0
            keep, self.hp = self.hp, self._hp_full  # (restored: capture_segments runs with the scaled copy)
            try:
                return base._add_l2_model(self, torch.zeros(1, dtype=torch.float32, device=self.device))
            finally:
                self.hp = keep

        # ---- hipGraph segments (fixed-capacity layout): the compute BETWEEN the collectives ----
        def capture_segments(self, idx, dense, y, mv=None):
            """Fixed-capacity layout only.  Captures the three compute stretches of every
            micro-batch - route | owner-side gather | embed..loss..backward..pack - as hipGraphs
            over static buffers; fwd_bwd then replays them with the RCCL calls issued eagerly in
            between (the whole step as ONE graph with the RCCL calls inside is bench.py's opt-in
            --graph-sharded: it could only be exercised at world size 1).  A step costs 3 graph launches + 3 collectives per micro-batch on the host
            instead of ~25 kernel launches.  idx / dense / y become the static input buffers:
            later calls with other tensors of the same shape are copied into them."""
            from types import SimpleNamespace

            from . import ops

            if not self.st.capacity_factor:
                raise ValueError("capture_segments needs the fixed-capacity exchange layout")
            M = max(1, self.micro_batches)
            B = idx.shape[0]
            if B % M:
                raise ValueError("batch must be divisible by micro_batches")
            b = B // M
            self._segs = None
            self._alloc(b)
            mvf = self._mv_fixed()
            if self.mv_fields and mv is None:
                raise ValueError("capture_segments: multi-valued / value features need their mv= entries")
            dev, W = self.device, self.D + PAD
            F = self.F_wide if mvf else self.F    # columns of the occurrence matrix the exchange carries
            extra = len(self.mv_fields) * b if mvf else 0  # pooled rows behind the received ones
            self._seg_wide = self.widen(idx, mv) if mvf else None
            cap = self.st.capacity(b * F)
            slots = world * cap
            router = self.st.route_fn
            router.ensure(b * F, world, cap)
            coll = world > 1 or (FORCE and dist.is_initialized())
            self.grad_scale = 1.0 / (world * M)
            # (the dense parameters' l2 shares are baked into the captured kernels: see fwd_bwd)
            self.hp = dict(self._hp_full, **{k: self._hp_full.get(k, 0.0) * self.grad_scale for k in _DENSE_L2})
            self._seg_loss = torch.zeros(1, dtype=torch.float32, device=dev)
            segs = []
            for c in range(M):
                s = SimpleNamespace(idx=idx[c * b: (c + 1) * b], dense=dense[c * b: (c + 1) * b],
                                    y=y[c * b: (c + 1) * b])
                if mvf:
                    s.iw, s.vw = (t[c * b: (c + 1) * b] for t in self._seg_wide)
                s.slots = slots
                s.pos = torch.empty(b * F, dtype=torch.int64, device=dev)
                s.send_ids = torch.empty(slots, dtype=torch.int64, device=dev)
                s.counts = torch.empty(world, dtype=torch.int64, device=dev)
                s.recv_ids = torch.empty_like(s.send_ids) if coll else s.send_ids
                served = torch.empty(slots + extra, W, dtype=torch.float32, device=dev)
                s.served = served[:slots]
                s.rows = torch.empty_like(served) if coll else served
                s.rows_recv = s.rows[:slots]
                s.grad_rows = torch.zeros(slots + extra, W, dtype=torch.float32, device=dev)
                s.grad_send = s.grad_rows[:slots]
                s.grad_out = torch.empty_like(s.grad_send) if coll else s.grad_send
                segs.append(s)

            def route(s, c):
                ops.shard_route_padded(s.iw if mvf else s.idx, self._foff_wide if mvf else self.field_off, world, cap,
                                       s.pos, s.send_ids, s.counts, router.overflow, router.ws)

            def gather(s, c):
                ops.gather_rows(self.st.shard[:, : self.st.W], s.recv_ids, s.served)

            def compute(s, c):
                self._slot = s
                self._wide = (s.iw, s.vw) if mvf else None
                try:
                    loss = self._fwd_bwd_packed(s.idx, s.dense, s.y, None, s.grad_rows, pos=s.pos)
                finally:
                    self._slot = self._wide = None
                if c == 0:
                    self._seg_loss.copy_(loss)
                else:
                    self._seg_loss.add_(loss)
                if M > 1:
                    if c == 0:
                        self._acc.copy_(self._flat_grads)
                    elif c < M - 1:
                        self._acc.add_(self._flat_grads)
                    else:
                        self._flat_grads.add_(self._acc)
                        self._seg_loss.div_(M)

            # eager rehearsal first (lazy workspaces, RCCL buffers), collectives included so the
            # static buffers hold real rows; then the captures, which only record
            self._segs, self._seg_coll, self._seg_in = segs, coll, (idx, dense, y)
            self._mv = mv
            self._seg_bodies = (route, gather, compute)
            self._run_segments(eager=True)
            torch.cuda.synchronize()
            import gc

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            # no cyclic garbage collection while a capture is open: the collector may free an OLDER
            # engine's graphs (they sit in reference cycles with the closures above), and releasing a
            # graph's memory pool is an unsafe call under torch's global capture mode - it aborted the
            # process when it happened (tests/test_gpu_dist.py run after other segment tests)
            gc.collect()
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.stream(side):
                    for c, s in enumerate(segs):
                        s.graphs = []
                        for body in (route, gather, compute):
                            g = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(g, stream=side):
                                body(s, c)
                            s.graphs.append(g)
            finally:
                if gc_was_on:
                    gc.enable()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.hp = self._hp_full

        def _run_segments(self, eager=False):
            S, coll = self._segs, self._seg_coll
            M = len(S)

            def run(s, c, k):
                if eager:
                    self._seg_bodies[k](s, c)
                else:
                    s.graphs[k].replay()

            def start(c):
                s = S[c]
                run(s, c, 0)
                if coll:
                    _all_to_all(s.recv_ids, s.send_ids, group=group)
                run(s, c, 1)
                return _all_to_all(s.rows_recv, s.served, group=group, async_op=True) if coll else None

            rows_work = start(0)
            works = []
            for c, s in enumerate(S):
                # the NEXT micro-batch's routing and row exchange go out before this one's compute
                nxt = start(c + 1) if c + 1 < M else None
                if rows_work is not None:
                    rows_work.wait()
                run(s, c, 2)
                if coll:
                    works.append(_all_to_all(s.grad_out, s.grad_send, group=group, async_op=True))
                rows_work = nxt
            for w in works:
                w.wait()
            if M == 1:
                self.shard_grad_ids, self.shard_grad_rows = S[0].recv_ids, S[0].grad_out
            else:
                self.shard_grad_ids = [s.recv_ids for s in S]
                self.shard_grad_rows = [s.grad_out for s in S]
            allreduce_dense(self.grads, world, group, self._flat_grads, average=False)
            return self._seg_loss + self._l2_once()

        def _replay_segments(self, idx, dense, y, mv=None):
            for src, dst in zip((idx, dense, y), self._seg_in):
                if src.shape != dst.shape:
                    raise ValueError("captured segments: batch shape differs from the captured one")
                if src.data_ptr() != dst.data_ptr():
                    dst.copy_(src)
            if self._seg_wide is not None:
                if mv is None:
                    raise ValueError("captured segments: multi-valued / value features need their mv= entries")
                self.widen(idx, mv, out=self._seg_wide)  # (the tags of THIS batch into the static wide buffers)
            return self._run_segments()

        def roofline_probes(self, idx, dense, y):
            """roofline_probe on the sharded engine: the LOCAL gather + FM + linear kernel over the
            rows received in the last exchange (no collective inside, any rank may call it)."""
            from . import ops

            if getattr(self, "ex", None) is None or getattr(self, "rows", None) is None:
                raise RuntimeError("run fwd_bwd once before roofline_probe")
            fm = self._has_fm()
            pos = self._pos                    # [b, F] of the (micro-)batch of the last exchange
            b = pos.shape[0]
            W = self.D + PAD
            flat = self.rows.view(-1)
            d = dense[:b].contiguous() if (self.use_linear and self.Dn) else None

            def fn():
                ops.embed_fwd(
                    pos, self.rows, self._zoff, table_ld=W, D=self.D,
                    bias_table=flat[self.D:] if fm else None, bias_ld=W,
                    lin_w=flat[self.D + 1:] if self.use_linear else None, lin_ld=W, lin_off=self._zoff,
                    lin_w_dense=self.linear_w_dense if (self.use_linear and self.Dn) else None,
                    lin_w0=self.params["linear_w0"] if self.use_linear else None, dense=d,
                    E=self.E[:b], fm_sum=self.fm_sum[:b] if fm else None,
                    fm_logit=self.fm_logit[:b] if fm else None,
                    lin_logit=self.lin_logit[:b] if self.use_linear else None)

            return [dict(name="embed_fwd_kernel on the exchanged rows (rm_embed_fwd: gather + FM + linear, "
                              f"{b} examples per launch)", symbol="embed_fwd_kernel", fn=fn,
                         work=self._embed_fwd_bytes(b, fm), bound="hbm")]

        def optimizer(self, name="adam", lr=1e-3):
            """The training step's second half for this engine: ShardedOptimizer(self, ...)."""
            return ShardedOptimizer(self, name, lr)

        def optimizer_probe_sharded(self, idx, dense, y, iters=5):
            """Times the owner-side optimizer step (every rank calls it: no collective inside, but the
            ranks stay in step).  Uses the gradients of the last fwd_bwd."""
            opt = self.optimizer("adam", 1e-3)
            for _ in range(2):
                opt.step()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(iters):
                opt.step()
            ev[1].record()
            torch.cuda.synchronize()
            n = sum(int(i.numel()) for i in (self.shard_grad_ids if isinstance(self.shard_grad_ids, list)
                                             else [self.shard_grad_ids]))
            return {"ms": round(ev[0].elapsed_time(ev[1]) / iters, 4), "gradient_rows_received": n,
                    "what": "owner-side row-wise lazy Adam on this rank's shard, straight from the gradient rows "
                            "the backward all_to_all delivered (rm_sparse_optimizer_step_rows), + Adam on the "
                            "all-reduced dense parameters (rm_dense_optimizer_step); NOT part of value"}

        def overflowed(self):
            """True when a fixed-capacity batch did not fit (host sync; clears the flag): every
            result since the last call must be discarded and redone with the dynamic layout."""
            flag = getattr(self.st.route_fn, "overflow", None)
            if flag is None or not self.st.capacity_factor:
                return False
            hit = bool(int(flag.item()))
            flag.zero_()
            return hit

    return Sharded()


class ShardedOptimizer:
    """The optimizer step of a row-sharded engine (what optimizer.minimize(...) over ALL variables is in
    the reference, xDeepFM.py:116-126, utils.py:201-213), owner side: every rank updates ITS shard rows
    straight from the gradient rows the backward all_to_all delivered (IndexedSlices: local row id +
    [dE | g_bias | g_lin | 0]; duplicates summed in arrival-independent order by the stable sort inside
    rm_sparse_optimizer_step_rows; empty fixed-capacity slots carry id -1 and are skipped), and its replica
    of the dense parameters from the all-reduced gradient (identical on every rank, so the replicas stay
    bit-identical).  Keras Adam / Adagrad / SGD, lazy on the table rows - the same semantics as
    optim.SparseTableOptimizer on one GPU.  The moments of a shard row's embedding live in a second
    [R_local, 2 D] array, those of its bias / linear entries in the row's own state columns."""

    def __init__(self, engine, name="adam", lr=1e-3):
        from . import ops
        from .optim import FusedDenseOptimizer

        if name not in ("adam", "adagrad", "gd", "sgd"):
            raise ValueError(f"ShardedOptimizer: {name!r} unsupported (adam, adagrad, sgd)")
        self.ops, self.e, self.name, self.lr = ops, engine, name, float(lr)
        st, D = engine.st, engine.D
        n = st.shard.shape[0]
        self.mom = None
        if name in ("adam", "adagrad"):
            self.mom = torch.zeros(n, 2 * D, device=st.shard.device)
            if name == "adagrad":
                self.mom.view(n, D // 4, 2, 4)[:, :, 1, :] = 0.1
        def init_state_columns():
            st.shard[:, D + 2: D + 6] = 0.0
            if name == "adagrad":
                st.shard[:, D + 4: D + 6] = 0.1
            if self.mom is not None:
                self.mom.zero_()
                if name == "adagrad":
                    self.mom.view(n, D // 4, 2, 4)[:, :, 1, :] = 0.1
            self.t = 0
            if getattr(self, "dense", None) is not None:
                self.dense.reset()

        self.t = 0
        init_state_columns()
        st.on_load = init_state_columns  # a restored shard starts the optimizer afresh (ShardedTable.load)
        self.dense = FusedDenseOptimizer(engine, name, lr)
        self._ws = None
        self.t = 0
        # embedding_l2_reg / linear_l2_reg, lazily: reg * row for the rows this step touches (once per row: the owner
        # adds it after summing every rank's gradient rows); the linear term's dense weights exactly
        hpf = getattr(engine, "_hp_full", engine.hp)
        self.l2_embedding = float(hpf.get("embedding_l2_reg", 0.0))
        self.l2_linear = float(hpf.get("linear_l2_reg", 0.0)) if engine.use_linear else 0.0

    def step(self, reset=False):
        e = self.e
        self.t += 1
        if self.l2_linear and e.Dn:
            e.grads["linear_w_dense"].add_(e.params["linear_w_dense"], alpha=self.l2_linear)
        ids, rows = e.shard_grad_ids, e.shard_grad_rows
        if isinstance(ids, list):  # micro-batches: ONE update per row from all their pieces together
            ids, rows = (ids[0], rows[0]) if len(ids) == 1 else (torch.cat(ids), torch.cat(rows))
        n = ids.numel()
        need = self.ops.sparse_optimizer_workspace(max(n, 1))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.zeros(need, dtype=torch.uint8, device=e.device)
        if n:
            self.ops.sparse_optimizer_step_rows(ids.contiguous(), rows.contiguous(), e.D, e.st.shard, self.mom,
                                                self._ws, self.t, self.name, self.lr, reset=reset,
                                                l2_embedding=self.l2_embedding, l2_linear=self.l2_linear)
        self.dense.step(reset=reset)
