"""Optimizer step for the engines' gradients (the step
either side of fwd+bwd).  Keras semantics of create_optimizer (recman/tf/core/utils.py:201-213):
Adam(beta .9/.999, epsilon 1e-7 outside the sqrt), Adagrad(initial accumulator 0.1,
epsilon 1e-7); plus plain SGD / momentum (the reference names for those do not exist in
tf.optimizers and raise).

Dense parameters are updated with torch foreach ops (plumbing).  Embedding-side
gradients arrive in IndexedSlices form (rows + idx); this version densifies them with
rm_scatter_add_rows - exactly what the reference's dense l2 term forces (layers.py:188-193) -
which is fine for ml-100k-sized tables.  A row-wise sparse update for Criteo-sized tables
is the next item."""
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


class SparseTableOptimizer:
    """Lazy row-wise update of the fused table rows straight from the IndexedSlices the
    backward produces (rm_sparse_optimizer_step): only rows occurring in the batch are
    touched, no dense gradient is ever formed.  Needs embedding_l2_reg == linear_l2_reg == 0
    (a dense l2 term touches every row, layers.py:188-193) - DeepModel falls back to the
    dense path otherwise.  Multi-valued (MultiValCsvFeat) and value (SparseValueFeat) features:
    their column of idx is masked out (-1) in the main call and every such feature is handed to
    the same kernel as an expanded one-field occurrence list - one occurrence per tag, its row
    gradient scaled by the pooling / value factor (rm_pool_rows_bwd's factors)."""

    def __init__(self, engine, name="adam", lr=1e-3):
        import torch as _t

        from . import ops

        self.ops, self.e, self.name, self.lr = ops, engine, name, float(lr)
        R, LD = engine.rows.shape
        dev = engine.device
        self.m = _t.zeros(R, LD, device=dev) if name == "adam" else None
        self.v = (_t.zeros(R, LD, device=dev) if name == "adam"
                  else _t.full((R, LD), 0.1, device=dev) if name == "adagrad" else None)
        self.gbuf = _t.zeros(R, LD, device=dev)
        self.stamp = _t.zeros(R, dtype=_t.int32, device=dev)
        self.t = 0

    def _lin_on(self, f):
        """Is embedding feature f among the linear features (hyper-parameter linear_features)?"""
        names = self.e.spec.linear_names
        return names is None or self.e.spec.sparse_names[f] in names

    def step(self, idx, reset=False):
        import torch as _t

        e = self.e
        self.t += 1
        g_bias = e.dlogit if (e.use_bias_tables and e._has_fm()) else None
        g_lin = e.dlogit if e.use_linear else None
        if e.mv_fields:
            idx = idx.clone()
            idx[:, e.mv_fields] = -1  # handled below
        self.ops.sparse_optimizer_step(
            idx, e.field_off, e.d_rows, e.rows, self.m, self.v, self.gbuf, self.stamp, self.t,
            self.name, self.lr, g_bias=g_bias, g_lin=g_lin, reset=reset,
            lin_field_mask=getattr(e, "lin_field_mask", None))
        B = idx.shape[0]
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
                ids.view(-1, 1).contiguous(), zoff, rows_g, e.rows, self.m, self.v, self.gbuf, self.stamp,
                self.t, self.name, self.lr,
                g_bias=(g_bias[seg] * wb).contiguous() if g_bias is not None else None,
                g_lin=(g_lin[seg] * wl).contiguous() if (g_lin is not None and self._lin_on(f)) else None,
                reset=reset)
