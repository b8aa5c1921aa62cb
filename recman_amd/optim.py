"""Optimizer step for the engines' gradients (the step
either side of fwd+bwd).  Keras semantics of create_optimizer (recman/tf/core/utils.py:201-213):
Adam(beta .9/.999, epsilon 1e-7 outside the sqrt), Adagrad(initial accumulator 0.1,
epsilon 1e-7); plus plain SGD / momentum (the reference names for those do not exist in
tf.optimizers and raise).

Three pieces:
  * Optimizer            dense parameters as a name -> tensor dict, torch foreach-style ops (plumbing;
                         also the small-table path: the embedding gradient densified by
                         rm_scatter_add_rows - what the reference's dense l2 term forces, layers.py:188-193);
  * FusedDenseOptimizer  the same rule over ONE flat buffer in one launch (rm_dense_optimizer_step):
                         the engine's dense parameters / gradients are re-homed as views of flat buffers;
  * SparseTableOptimizer row-wise LAZY step on the table rows a batch touched (rm_sparse_optimizer_step):
                         stable sort by row, duplicates summed in occurrence order, parameter row + moments
                         updated in one pass - deterministic, no float atomics, no dense gradient.
                         LazyAdam is a deliberate deviation from Keras' sparse Adam (which decays the
                         moments of every row each step): identical under the reference's
                         new-optimizer-per-batch quirk (reset=True) and when every row is touched every
                         step; DESIGN.md section 6, tests/test_gpu_optim.py."""
import math

import torch


class Optimizer:
    def __init__(self, name="adam", lr=1e-3):
        if name not in ("adam", "adagrad", "gd", "sgd", "momentum"):
            raise ValueError(f"unknown optimizer {name!r}")  # utils.py:213
        self.name, self.lr = name, float(lr)
        self.t = 0
        self.state = {}

    def reset(self):
        """Forget every moment: the reference builds a NEW optimizer for every batch
        (xDeepFM.py:121-126); strict_reference mode calls this before each step."""
        self.t = 0
        self.state = {}

    @torch.no_grad()
    def step(self, params, grads):
        """params / grads: name -> tensor (same shapes).  In-place update."""
        self.t += 1
        for k, g in grads.items():
            p = params[k]
            if self.name == "adam":
                st = self.state.setdefault(k, (torch.zeros_like(p), torch.zeros_like(p)))
                m, v = st
                m.mul_(0.9).add_(g, alpha=0.1)
                v.mul_(0.999).addcmul_(g, g, value=0.001)
                lr_t = self.lr * math.sqrt(1 - 0.999 ** self.t) / (1 - 0.9 ** self.t)
                p.addcdiv_(m, v.sqrt().add_(1e-7), value=-lr_t)
            elif self.name == "adagrad":
                acc = self.state.setdefault(k, torch.full_like(p, 0.1))
                acc.addcmul_(g, g)
                p.addcdiv_(g, acc.sqrt().add_(1e-7), value=-self.lr)
            elif self.name == "momentum":
                buf = self.state.setdefault(k, torch.zeros_like(p))
                buf.mul_(0.9).add_(g)
                p.add_(buf, alpha=-self.lr)
            else:
                p.add_(g, alpha=-self.lr)


def flatten_views(d, keys):
    """Re-homes d[k] (k in keys) as views of ONE flat buffer (16-byte aligned slots); the values are
    kept.  Mutates d; returns the flat buffer."""
    offs, n = [], 0
    for k in keys:
        offs.append(n)
        n += (d[k].numel() + 3) // 4 * 4
    any_t = d[keys[0]]
    flat = torch.zeros(n, dtype=any_t.dtype, device=any_t.device)
    for k, o in zip(keys, offs):
        v = flat[o: o + d[k].numel()].view(d[k].shape)
        v.copy_(d[k])
        d[k] = v
    return flat


class FusedDenseOptimizer:
    """Dense parameters of an engine in ONE launch per step.  Construction re-homes the engine's dense
    parameters and their gradients as views of two flat buffers (same values; the kernels read the
    dicts at every call, so nothing else changes - but build it BEFORE capturing a hipGraph of the
    step).  Adam / Adagrad / SGD with the Keras constants."""

    def __init__(self, engine, name="adam", lr=1e-3):
        from . import ops

        if name not in ("adam", "adagrad", "gd", "sgd"):
            raise ValueError(f"FusedDenseOptimizer: {name!r} unsupported (adam, adagrad, sgd)")
        self.ops, self.e, self.name, self.lr = ops, engine, name, float(lr)
        flat_g = getattr(engine, "_flat_grads", None)  # the row-sharded engines flattened theirs already
        self.keys = sorted(engine.grads)
        if flat_g is None:
            flat_g = flatten_views(engine.grads, self.keys)
            engine._flat_grads = flat_g
        self.g = flat_g
        self.p = flatten_views(engine.params, self.keys)
        if "linear_w_dense" in engine.params:
            engine.linear_w_dense = engine.params["linear_w_dense"]  # (the engines also hold it as an attribute)
        assert self.p.numel() == self.g.numel()
        self.m = torch.zeros_like(self.p) if name == "adam" else None
        self.v = (torch.zeros_like(self.p) if name == "adam"
                  else torch.full_like(self.p, 0.1) if name == "adagrad" else None)
        self.t = 0

    def reset(self):
        self.t = 0
        if self.m is not None:
            self.m.zero_()
        if self.v is not None:
            self.v.fill_(0.0 if self.name == "adam" else 0.1)

    def step(self, params=None, grads=None, reset=False):
        """(params / grads are accepted for interface parity with Optimizer.step and ignored: the
        engine's own flat buffers are updated.)"""
        self.t += 1
        self.ops.dense_optimizer_step(self.p, self.g, self.m, self.v, self.t, self.name, self.lr, reset=reset)


class SparseTableOptimizer:
    """Lazy row-wise update of the fused table rows straight from the IndexedSlices the
    backward produces (rm_sparse_optimizer_step): only rows occurring in the batch are
    touched, no dense gradient is ever formed, results are bit-reproducible.  embedding_l2_reg /
    linear_l2_reg are applied LAZILY (l2_embedding / l2_linear: reg * row added to the gradient of the rows a
    batch touches, once per distinct row and step; the reference's dense term touches every row every step,
    layers.py:188-193 - the two coincide when every row is touched each step).  Needs a
    fused row with room for the four moment
    entries of its bias / linear columns (LD >= D + 8: D >= 8 with the engines' LD = 2 D rows).
    State: those four entries in the row itself + one [R, 2 D] array of the embedding entries' moments,
    interleaved [m4 v4] per float4 slice (moments(): the plain [R, D] views).
    Multi-valued (MultiValCsvFeat) and value (SparseValueFeat) features: their column of idx is
    masked out (-1) in the main call and every such feature is handed to the same kernel as an
    expanded one-field occurrence list - one occurrence per tag, its row gradient scaled by the
    pooling / value factor (rm_pool_rows_bwd's factors)."""

    def __init__(self, engine, name="adam", lr=1e-3, l2_embedding=0.0, l2_linear=0.0):
        from . import ops

        if name not in ("adam", "adagrad", "gd", "sgd"):
            raise ValueError(f"SparseTableOptimizer: {name!r} unsupported (adam, adagrad, sgd)")
        self.ops, self.e, self.name, self.lr = ops, engine, name, float(lr)
        self.l2_embedding, self.l2_linear = float(l2_embedding), float(l2_linear)
        R, LD = engine.rows.shape
        D = engine.D
        if LD < D + 8 or not 8 <= D <= 64:
            raise ValueError(f"SparseTableOptimizer needs 8 <= D <= 64 and table rows of at least D + 8 floats "
                             f"(D={D}, LD={LD}): use the dense optimizer for this table")
        dev = engine.device
        self.D = D
        self.mom = None
        if name == "adam":
            self.mom = torch.zeros(R, 2 * D, device=dev)
        elif name == "adagrad":
            self.mom = torch.zeros(R, 2 * D, device=dev)
            self.mom.view(R, D // 4, 2, 4)[:, :, 1, :] = 0.1  # the v halves ([m4 v4] per float4 slice)
        # the bias / linear columns' moments live in the row's padding: [.. bias lin | m_b m_l v_b v_l | ..]
        engine.rows[:, D + 2: D + 6] = 0.0
        if name == "adagrad":
            engine.rows[:, D + 4: D + 6] = 0.1
        self._ws = None
        self.t = 0

    def moments(self):
        """(m [R, D], v [R, D]) of the embedding entries as strided views of the interleaved state."""
        R = self.mom.shape[0]
        q = self.mom.view(R, self.D // 4, 2, 4)
        return q[:, :, 0, :].reshape(R, self.D), q[:, :, 1, :].reshape(R, self.D)

    def _workspace(self, n):
        need = self.ops.sparse_optimizer_workspace(n)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.zeros(need, dtype=torch.uint8, device=self.e.device)  # (zeroed: it carries counters)
        return self._ws

    def _lin_on(self, f):
        """Is embedding feature f among the linear features (hyper-parameter linear_features)?"""
        names = self.e.spec.linear_names
        return names is None or self.e.spec.sparse_names[f] in names

    def _max_field_rows(self):
        """The engine's fields own consecutive row ranges of feat_sizes[f] rows (engine.py, field_off = their running
        sum): the step may sort the ids per field (rm_sparse_optimizer_step, max_field_rows).  Hyper-parameter
        optimizer_sort = "rows" keeps the one sort over all (row, occurrence) pairs."""
        if self.e.hp.get("optimizer_sort", "fields") != "fields" or len(self.e.spec.feat_sizes) > 64:
            return 0
        return int(max(self.e.spec.feat_sizes))

    def prepare(self, idx):
        """The id-only part of the next step (keys + stable sort by row) on the CURRENT stream: a third
        of a step's time that needs no gradient - fit() issues it on a side stream beside the
        forward+backward pass.  The next step(idx) on the same ids skips it."""
        e = self.e
        if e.spec.scratch_names:
            return  # (the expanded occurrence lists of multi-valued features are built in step())
        self.ops.sparse_optimizer_prepare(self._workspace(idx.numel()), e.rows.shape[0], idx=idx,
                                          field_off=e.field_off, max_field_rows=self._max_field_rows())
        # (the sort is tied to the tensor's CONTENT: data pointer, shape and torch's version counter, which every
        # in-place write - a feeder refilling a static input buffer - advances)
        self._prepared = (idx.data_ptr(), tuple(idx.shape), idx._version)

    def step(self, idx, reset=False):
        import torch as _t

        e = self.e
        self.t += 1
        prepared = (getattr(self, "_prepared", None) == (idx.data_ptr(), tuple(idx.shape), idx._version)
                    and not e.mv_fields)
        self._prepared = None
        g_bias = e.dlogit if (e.use_bias_tables and e._has_fm()) else None
        g_lin = e.dlogit if e.use_linear else None
        if e.mv_fields:
            idx = idx.clone()
            idx[:, e.mv_fields] = -1  # handled below
        B, F = idx.shape
        self.ops.sparse_optimizer_step(
            idx, e.field_off, e.d_rows, e.rows, self.mom, self._workspace(B * F), self.t,
            self.name, self.lr, g_bias=g_bias, g_lin=g_lin, reset=reset,
            lin_field_mask=getattr(e, "lin_field_mask", None), prepared=prepared,
            l2_embedding=self.l2_embedding, l2_linear=self.l2_linear if e.use_linear else 0.0,
            max_field_rows=self._max_field_rows())
        for f in e.mv_fields:
            offsets, ids, vals = e._mv_entry(f)
            n = offsets[1:] - offsets[:-1]
            seg = _t.repeat_interleave(_t.arange(B, device=ids.device), n)
            if vals is not None:  # value feature: emb and linear scaled by the value, bias not
                we, wb, wl = vals, _t.ones_like(vals), vals
            else:                 # sqrtn combiner; the linear term is a multi-hot count without slot 0
                inv = n.clamp(min=1).to(_t.float32).rsqrt()[seg]
                we, wb, wl = inv, inv, (ids >= 1).to(_t.float32)
            rows_g = (e.d_rows[seg, f, :] * we.unsqueeze(1)).contiguous().view(-1, 1, e.D)
            zoff = e.field_off[f: f + 1]
            self.ops.sparse_optimizer_step(
                ids.view(-1, 1).contiguous(), zoff, rows_g, e.rows, self.mom, self._workspace(ids.numel()),
                self.t, self.name, self.lr,
                g_bias=(g_bias[seg] * wb).contiguous() if g_bias is not None else None,
                g_lin=(g_lin[seg] * wl).contiguous() if (g_lin is not None and self._lin_on(f)) else None,
                reset=reset, l2_embedding=self.l2_embedding,
                l2_linear=self.l2_linear if (e.use_linear and self._lin_on(f)) else 0.0)

    def roofline(self, idx, ms):
        """The step against the HBM roofline: bytes a step HAS to move with this layout - per occurrence
        its index and gradient row, per DISTINCT row the parameter line and the moment line read and
        written - over the measured time."""
        e, D = self.e, self.D
        n = idx.numel()
        distinct = int(torch.unique((idx + e.field_off).reshape(-1)).numel())
        ld_bytes = e.rows.shape[1] * 4
        per_row = 2 * ld_bytes + (2 * 2 * D * 4 if self.mom is not None else 0)
        work = n * (8 + 4 * D + 8) + distinct * per_row
        gbs = work / (ms * 1e-3) / 1e9
        return {"kernel": "rm_sparse_optimizer_step (keys + field-segmented radix sort + sparse_apply_kernel)", "bound": "hbm",
                "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                "algorithmic_per_step": work, "occurrences": n, "distinct_rows": distinct,
                "bytes": "per occurrence idx 8 + gradient row 4D + sorted pair 8; per distinct row "
                         f"{ld_bytes} B parameter line + {2 * D * 4} B moment line, read and written"}
