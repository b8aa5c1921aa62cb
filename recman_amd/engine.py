"""Explicit forward+backward engines for DeepFM / DCN / xDeepFM on one MI355X.

No autograd and no tracing compiler: each engine owns its parameters (laid out
for the kernels: ONE concatenated embedding table in HBM, per-feature variables
are views of it), a workspace sized once per batch size, and enqueues a fixed
sequence of HIP kernels (recman_amd/ops.py -> librecman_hip.so) on the current
stream - which makes a whole step capturable in a hipGraph (`capture=True`).

The MLP ("DNN", layers.py:576-609) runs on hand-written f32-MFMA kernels as well:
csrc/mlp.hip (all layers fused, hidden widths <= 32) or csrc/gemm.hip (one launch per
GEMM with fused epilogues, any width); no library GEMM is left on the path.

Variable names are the reference's (layers.py:96,106,318,324,533,541,548,558,564,
572,663,673,687,693) so state_dict() round-trips with its checkpoints' keys.
"""
import torch

from . import ops

F32, I64 = torch.float32, torch.int64

_ACTS = {"relu": 0.0, "leaky_relu": 0.2}


def act_name(a):
    """Maps the reference's activation argument (a TF callable such as tf.nn.relu,
    DeepFM.py:40, or a name) to a name this package knows."""
    if a is None:
        return "identity"
    if isinstance(a, str):
        name = a
    else:
        name = getattr(a, "__name__", str(a))
    name = name.lower()
    for k in ("leaky_relu", "relu", "identity", "linear"):
        if k in name:
            return "identity" if k == "linear" else k
    raise ValueError(f"unsupported activation {a!r} (relu, leaky_relu, identity)")


class FeatureSpec:
    """Host-side description of the inputs: embedding (sparse) features in
    FeatureDictionary order (inputs.py:13-15) with their feat_size (null slot
    included, inputs.py:166) and the dense feature names."""

    def __init__(self, sparse_names, feat_sizes, dense_names=(), multi_names=(), value_names=(),
                 linear_names=None):
        self.sparse_names = list(sparse_names)
        self.feat_sizes = [int(v) for v in feat_sizes]
        self.dense_names = list(dense_names)
        # the hyper-parameter linear_features (get_linear_features, utils.py:27-30): the features of
        # the linear term in the order given; None = every feature in the default order (:31-36)
        self.linear_names = list(linear_names) if linear_names else None
        if self.linear_names is not None:
            known = set(self.sparse_names) | set(self.dense_names)
            bad = [n for n in self.linear_names if n not in known]
            if bad or len(set(self.linear_names)) != len(self.linear_names):
                raise ValueError(f"linear_features: unknown or repeated features {bad or self.linear_names}")
        # embedding features that are multi-valued (MultiValCsvFeat): sqrtn-pooled lookup
        self.multi_names = list(multi_names)
        # embedding features that carry a value (SparseValueFeat): value-weighted lookup
        self.value_names = list(value_names)
        if len(self.sparse_names) != len(self.feat_sizes):
            raise ValueError("sparse_names and feat_sizes differ in length")

    @property
    def F(self):
        return len(self.sparse_names)

    @property
    def Dn(self):
        return len(self.dense_names)

    @property
    def rows(self):
        return sum(self.feat_sizes)

    def offsets(self):
        off, out = 0, []
        for v in self.feat_sizes:
            out.append(off)
            off += v
        return out

    def lin_ref_blocks(self):
        """(row offset in the table, size) of each feature's one-hot block in the order the
        reference's linear_w stacks them (utils.py:27-36): sparse feats, value feats, then
        multi-valued."""
        at = dict(zip(self.sparse_names, zip(self.offsets(), self.feat_sizes)))
        special = set(self.multi_names) | set(self.value_names)
        order = ([n for n in self.sparse_names if n not in special]
                 + [n for n in self.sparse_names if n in self.value_names]
                 + [n for n in self.sparse_names if n in self.multi_names])
        return [at[n] for n in order]

    def lin_ref_layout(self):
        """The reference's linear_w as a list of pieces in ITS order: ("s", first table row, size)
        for an embedding feature's one-hot block, ("d", j) for dense column j."""
        if self.linear_names is None:
            return [("s", o, n) for o, n in self.lin_ref_blocks()] + [("d", j) for j in range(self.Dn)]
        at = dict(zip(self.sparse_names, zip(self.offsets(), self.feat_sizes)))
        dj = {n: j for j, n in enumerate(self.dense_names)}
        return [("s",) + at[n] if n in at else ("d", dj[n]) for n in self.linear_names]

    def lin_masks(self):
        """(per-field 0/1 list, per-dense-column 0/1 list): which features the linear term uses;
        (None, None) when it uses all of them."""
        if self.linear_names is None:
            return None, None
        sel = set(self.linear_names)
        return ([1.0 if n in sel else 0.0 for n in self.sparse_names],
                [1.0 if n in sel else 0.0 for n in self.dense_names])

    @property
    def scratch_names(self):
        """Features whose per-example row is computed into scratch rows before the gather."""
        return self.multi_names + self.value_names


class MLP:
    """DNN.__call__ (layers.py:576-609) with an explicit backward.  x = [xe | xd] is
    never concatenated: layer 0 is two GEMMs accumulating into one output."""

    def __init__(self, params, grads, FD, Dn, hidden, activation, device, prefix=""):
        self.FD, self.Dn = FD, Dn
        self.hidden = list(hidden)
        self.act = act_name(activation)
        if self.act not in ("relu", "leaky_relu", "identity"):
            raise ValueError(self.act)
        self.p, self.g, self.prefix = params, grads, prefix
        dims = [FD + Dn] + self.hidden
        for i in range(len(self.hidden)):
            for nm, shape in ((f"{prefix}dnn_layer_{i}_weights", (dims[i], dims[i + 1])),
                              (f"{prefix}dnn_layer_{i}_bias", (dims[i + 1],))):
                params[nm] = torch.zeros(shape, dtype=F32, device=device)
                grads[nm] = torch.zeros(shape, dtype=F32, device=device)
        for nm, shape in ((f"{prefix}dnn_w", (dims[-1], 1)), (f"{prefix}dnn_w0", (1,))):
            params[nm] = torch.zeros(shape, dtype=F32, device=device)
            grads[nm] = torch.zeros(shape, dtype=F32, device=device)
        self._B = None
        self._ws = None
        self._ones = None
        # hand-written fused f32-MFMA path for skinny MLPs (csrc/mlp.hip); wider ones (DCN's
        # [400,400]) and any MLP under dropout run layer by layer on the wide dense kernels
        # (csrc/gemm.hip: bias / activation / activation-gradient fused, x = [xe | xd] in place)
        self.fused_ok = bool(FD % 4 == 0 and ops.mlp_supported(FD, Dn, self.hidden))
        self.fused = False
        # hp["d_rows_reuse"] = "stream": the row gradients leave the caches (non-temporal stores) - only
        # for a bare forward+backward; the default keeps them cached for the optimizer step that
        # gathers them right after (fit(), and the benchmark: it times what training runs)
        self.stream_d_rows = False

    def _alloc(self, B, device):
        if self._B == B:
            return
        self._B = B
        if self.fused_ok:
            self.hb = [torch.zeros(B, 32, dtype=F32, device=device) for _ in self.hidden]
            self.dhb = [torch.zeros(B, 32, dtype=F32, device=device) for _ in self.hidden]
            self.fws = torch.empty(ops.mlp_bwd_workspace(self.FD, self.Dn), dtype=F32, device=device)
            self.ones = torch.ones(B, dtype=F32, device=device)
            self.tmp32 = torch.empty(32, dtype=F32, device=device)
        self.a = [torch.empty(B, h, dtype=F32, device=device) for h in self.hidden]
        self.da = [torch.empty(B, h, dtype=F32, device=device) for h in self.hidden]
        self.out = torch.empty(B, 1, dtype=F32, device=device)

    def _act_(self, h):
        if self.act == "relu":
            torch.relu_(h)
        elif self.act == "leaky_relu":
            torch.nn.functional.leaky_relu_(h, 0.2)
        return h

    def forward(self, xe, xd, keep=None, masks=None, head=None):
        """xe [B,FD], xd [B,Dn] or None -> logit [B] (a view of an internal buffer).
        keep/masks: DNN dropout keep-probabilities and 0/1 masks (layers.py:589,602).
        head: keyword arguments of ops.mlp_tail (minus dh) - when the fused kernel runs, the final
        logit, prediction, loss and dLoss/dlogit are produced in its epilogue together with the dh
        chain (self.head_done tells the caller; otherwise it runs rm_logit_loss as usual)."""
        B = xe.shape[0]
        self._alloc(B, xe.device)
        p, pre = self.p, self.prefix
        n = len(self.hidden)
        self.keep = keep if keep is not None else [1] * (n + 1)
        self.masks = masks if masks is not None else [None] * (n + 1)
        self.xe, self.xd = xe, xd
        dropping = any(k < 1 and mk is not None for k, mk in zip(self.keep, self.masks))
        self.fused = self.fused_ok and not dropping
        self.head_done, self.tail = False, None
        if self.fused:
            if head is not None:
                self.tail = ops.mlp_tail(B, dh=self.dhb, **head)
                self.head_done = True
            ops.mlp_fwd(xe, xd if self.Dn else None,
                        [p[f"{pre}dnn_layer_{i}_weights"] for i in range(n)],
                        [p[f"{pre}dnn_layer_{i}_bias"] for i in range(n)], p[f"{pre}dnn_w"].view(-1),
                        p[f"{pre}dnn_w0"], self.act, self.hb, self.out.view(B), tail=self.tail)
            return self.out.view(B)
        if self.keep[0] < 1 and self.masks[0] is not None:
            m = self.masks[0] / self.keep[0]
            xe = xe * m[:, : self.FD]
            xd = xd * m[:, self.FD:] if xd is not None else None
            self.xe, self.xd = xe, xd
        self._alloc_dense(xe.device)
        dotted = False
        for i in range(n):
            W, b = p[f"{pre}dnn_layer_{i}_weights"], p[f"{pre}dnn_layer_{i}_bias"]
            a = self.a[i]
            if i == 0:
                ops.dense_fwd(xe, xd if self.Dn else None, W, a, self._fws, bias=b, act=self.act, ws6=self._fws6)
            else:
                # the last layer's kernel also forms the output projection a . dnn_w + dnn_w0 from its registers
                dot = None
                if i == n - 1 and not (self.keep[n] < 1 and self.masks[n] is not None):
                    dot = (p[f"{pre}dnn_w"].view(-1), p[f"{pre}dnn_w0"], self.out.view(B))
                dotted = ops.dense_fwd(self.a[i - 1], None, W, a, self._fws, bias=b, act=self.act, ws6=self._fws6,
                                       dot=dot) and dot is not None
            if self.keep[i + 1] < 1 and self.masks[i + 1] is not None:
                a.mul_(self.masks[i + 1] / self.keep[i + 1])
        if not dotted:
            ops.rowdot(self.a[-1], p[f"{pre}dnn_w"].view(-1), p[f"{pre}dnn_w0"], self.out.view(B))
        return self.out.view(B)

    def _alloc_dense(self, device):
        if getattr(self, "_fws", None) is not None and self._fws.device == device and self._wws_B == self._B:
            return
        dims = [self.FD + self.Dn] + self.hidden
        fw = max(ops.dense_filter_workspace(max(dims[i], dims[i + 1]), max(dims[i], dims[i + 1]))
                 for i in range(len(self.hidden)))
        ww = max(ops.dense_wgrad_workspace(dims[i], dims[i + 1], self._B) for i in range(len(self.hidden)))
        self._fws = torch.empty(fw, dtype=F32, device=device)
        # wide layers on the bf16 matrix pipe with split fp32 operands (rm_dense_fwd6, csrc/gemm6.hip) unless
        # dense_gemm = "f32" asks for the f32 MFMA kernel
        self._fws6 = self._wws6 = None
        if getattr(self, "dense_gemm", "bf16x6") == "bf16x6":
            kmax = max(dims)
            self._fws6 = torch.empty(ops.dense6_workspace(kmax, kmax, self._B), dtype=F32, device=device)
            # (+ 16 columns: DCN rides the cross net's coefficient columns along the first layer's pass, wgrad0)
            self._wws6 = torch.empty(max(ops.dense_wgrad6_workspace(dims[i], dims[i + 1] + 16, self._B)
                                         for i in range(len(self.hidden))), dtype=F32, device=device)
        self._wws = torch.empty(max(ww, 1), dtype=F32, device=device)
        self._wws_B = self._B
        self._ws = torch.empty(256 * 1024, dtype=F32, device=device)
        self._ones = torch.ones(self._B, dtype=F32, device=device)

    def wgrad0(self, G2=None, dW2=None):
        """The first layer's weight gradient of a backward(..., defer_wgrad0=True) call, optionally with a second piece
        of gradient columns G2 [B, N2] -> dW2 [K, N2] against the SAME x = [xe | xd] in the same pass (DCN: the cross
        net's coefficient columns - x0 is then read once, rm_dense_wgrad6)."""
        da, gW, db = self._deferred
        self._deferred = None
        ops.dense_wgrad(self.xe, self.xd if self.Dn else None, da, gW, self._wws, db=db, ws6=self._wws6, G2=G2, dW2=dW2)

    def can_defer_wgrad0(self):
        """backward(defer_wgrad0=True) + wgrad0(G2, dW2) is available: wide layers on the split-operand path, no input
        dropout (the MLP's x is then the caller's x0 itself)."""
        return (not self.fused_ok and getattr(self, "_wws6", None) is not None
                and not (self.keep[0] < 1 and self.masks[0] is not None))

    def backward(self, g, dxe, fm_sum=None, lin_grads=None, defer_wgrad0=False):
        """g [B] = dLoss/dlogit; writes dLoss/dxe into dxe [B,FD] and the parameter
        gradients into self.g.  (No gradient is needed for the dense inputs.)
        fm_sum [B,D]: also add the FM second-order gradient g*(S - E) (fused path only;
        returns True when it was added).  lin_grads = (d_w_dense [Dn], d_w0 [1]): the linear
        term's dense-weight gradients g^T xd and sum g ride along too (fused path, Dn <= 32):
        self.lin_done tells the caller whether they were written."""
        p, gr, pre = self.p, self.g, self.prefix
        n = len(self.hidden)
        self.lin_done = False
        if self.fused:
            Ws = [p[f"{pre}dnn_layer_{i}_weights"] for i in range(n)]
            ops.mlp_bwd(self.xe, self.xd if self.Dn else None, Ws, p[f"{pre}dnn_w"].view(-1), self.act,
                        g, self.hb, dxe, self.dhb, [gr[f"{pre}dnn_layer_{i}_weights"] for i in range(n)],
                        self.fws, fm_sum=fm_sum,
                        db=[gr[f"{pre}dnn_layer_{i}_bias"] for i in range(n)],
                        d_w_out=gr[f"{pre}dnn_w"].view(-1), d_w0_out=gr[f"{pre}dnn_w0"],
                        d_xd_wsum=lin_grads[0] if (lin_grads and 1 <= self.Dn <= 32) else None,
                        d_g_sum=lin_grads[1] if (lin_grads and 1 <= self.Dn <= 32) else None,
                        tail=self.tail if self.head_done else None,
                        stream_d_rows=getattr(self, "stream_d_rows", False))
            self.lin_done = bool(lin_grads and 1 <= self.Dn <= 32)
            return fm_sum is not None
        # d(pre-activation of the last layer) = (g w_out^T) o mask o act'(a): one elementwise pass
        # (rm_outer_actgrad; widths that are not a multiple of 4 take the K = 1 GEMM instead)
        da = self.da[-1]
        last_drop = self.keep[n] < 1 and self.masks[n] is not None
        plain = last_drop or self.act == "identity"
        # ... which, without dropout on that layer, also reduces the columns it reads and writes:
        # d dnn_w = a^T g, d dnn_w0 = sum g and the last hidden layer's bias gradient
        sums = (not plain) and da.shape[1] % 4 == 0 and da.shape[1] >= 64
        if sums:
            need = ops.outer_actgrad_sums_workspace(da.shape[0], da.shape[1])
            if getattr(self, "_sums_ws", None) is None or self._sums_ws.numel() < need:
                self._sums_ws = torch.empty(need, dtype=F32, device=da.device)
            ops.outer_actgrad_sums(g, p[f"{pre}dnn_w"].view(-1), self.a[-1], self.act, da,
                                   gr[f"{pre}dnn_w"].view(-1), gr[f"{pre}dnn_w0"],
                                   gr[f"{pre}dnn_layer_{n - 1}_bias"], self._sums_ws)
        else:
            ops.linear_dense_bwd(g, self.a[-1], gr[f"{pre}dnn_w"].view(-1), gr[f"{pre}dnn_w0"], self._ws)
        if sums:
            pass
        elif da.shape[1] % 4 == 0:
            ops.outer_actgrad(g, p[f"{pre}dnn_w"].view(-1), None if plain else self.a[-1], self.act, da)
        else:
            ops.dense_fwd(g.view(-1, 1), None, p[f"{pre}dnn_w"], da, self._fws, transposed=True,
                          epilogue=ops.DENSE_ADD if plain else ops.DENSE_MUL_ACTGRAD,
                          act=self.act, aux1=None if plain else self.a[-1])
        if last_drop:
            da.mul_(self.masks[n] / self.keep[n])
            if self.act != "identity":
                ops.act_bwd_(da, self.a[-1], self.act) if da.numel() % 4 == 0 else da.mul_(
                    torch.where(self.a[-1] > 0, 1.0, _ACTS[self.act] or 0.0))
        for i in range(n - 1, -1, -1):
            W = p[f"{pre}dnn_layer_{i}_weights"]
            gW = gr[f"{pre}dnn_layer_{i}_weights"]
            # the bias gradient colsum(da): from rm_outer_actgrad_sums for the last layer, otherwise
            # it rides along in the weight-gradient kernel (which stages da in LDS anyway)
            db = None if (sums and i == n - 1) else gr[f"{pre}dnn_layer_{i}_bias"]
            if i == 0:
                if defer_wgrad0:
                    self._deferred = (da, gW, db)
                else:
                    ops.dense_wgrad(self.xe, self.xd if self.Dn else None, da, gW, self._wws, db=db, ws6=self._wws6)
                # dLoss/dxe = da W[:FD]^T (the dense inputs need no gradient)
                ops.dense_fwd(da, None, W[: self.FD], dxe, self._fws, transposed=True, epilogue=ops.DENSE_ADD,
                              ws6=self._fws6)
                if self.keep[0] < 1 and self.masks[0] is not None:
                    dxe.mul_(self.masks[0][:, : self.FD] / self.keep[0])
            else:
                prev = self.a[i - 1]
                ops.dense_wgrad(prev, None, da, gW, self._wws, db=db, ws6=self._wws6)
                dropped = self.keep[i] < 1 and self.masks[i] is not None
                # d(pre-activation of layer i-1) = (da W^T) o mask o act'(prev), act' from the stored
                # post-activation values (dropped positions are zeroed by the mask)
                ops.dense_fwd(da, None, W, self.da[i - 1], self._fws, transposed=True,
                              epilogue=ops.DENSE_MUL_ACTGRAD if self.act != "identity" else ops.DENSE_ADD,
                              act=self.act, aux1=prev if self.act != "identity" else None, ws6=self._fws6)
                da = self.da[i - 1]
                if dropped:
                    da.mul_(self.masks[i] / self.keep[i])
        return False

    def l2(self, reg):
        ws = [self.p[f"{self.prefix}dnn_layer_{i}_weights"] for i in range(len(self.hidden))]
        ws.append(self.p[f"{self.prefix}dnn_w"])
        return sum(reg * 0.5 * w.square().sum() for w in ws)  # layers.py:611-628

    def add_l2_grads(self, reg):
        for i in range(len(self.hidden)):
            nm = f"{self.prefix}dnn_layer_{i}_weights"
            self.g[nm].add_(self.p[nm], alpha=reg)
        nm = f"{self.prefix}dnn_w"
        self.g[nm].add_(self.p[nm], alpha=reg)


def init_reference(engine, seed=2019):
    """Initial values with the reference's distributions (TF's RNG stream itself cannot
    be reproduced): embedding tables, DNN and CIN weights truncated-normal glorot
    (utils.py:180-183; layers.py:99-101,536,551,567,666), cin_w glorot-uniform
    (layers.py:690), everything else zeros (layers.py:109,321,327,544,561,574,676,695).
    Cross-net vectors (absent from the reference): glorot-normal weights, zero biases."""
    import math

    g = torch.Generator(device=engine.device).manual_seed(int(seed))

    def tn(t, fan_in, fan_out):
        std = math.sqrt(2.0 / (fan_in + fan_out))
        torch.nn.init.trunc_normal_(t, 0.0, std, -2 * std, 2 * std, generator=g)

    for name, t in engine.params.items():
        if name.endswith("_feat_embed") or name.endswith("_weights") or name == "dnn_w":
            tn(t, t.shape[0], t.shape[1])
        elif name.startswith("cin_filter_"):
            tn(t, t.shape[1], t.shape[2])
        elif name == "cin_w":
            b = math.sqrt(6.0 / (t.shape[0] + t.shape[1]))
            t.uniform_(-b, b, generator=g)
        elif name in ("cross_w",):
            tn(t, t.shape[1], t.shape[2] if t.dim() == 3 else 1)
        elif name == "cross_w_out":
            tn(t, t.shape[0], 1)
        else:
            t.zero_()


class Engine:
    """Shared storage + the embedding / linear / loss plumbing.  Subclasses add the
    model-specific branches and define `_branches_fwd` / `_branches_bwd`."""

    model = "base"
    use_bias_tables = False

    def __init__(self, spec, embedding_size, hp, task="classification", device="cuda"):
        if not torch.cuda.is_available():
            raise RuntimeError("recman_amd engines need a GPU (MI355X); there is no CPU path")
        self.spec, self.D, self.hp, self.task = spec, int(embedding_size), dict(hp), task
        self.device = torch.device(device)
        dev = self.device
        F, R, Dn = spec.F, spec.rows, spec.Dn
        self.F, self.Dn, self.FD = F, Dn, F * self.D
        self.params, self.grads = {}, {}
        self._alloc_tables()
        self.params["linear_w0"] = torch.zeros(1, dtype=F32, device=dev)
        self.grads["linear_w0"] = torch.zeros(1, dtype=F32, device=dev)
        self.grads["linear_w_dense"] = torch.zeros(Dn, dtype=F32, device=dev)
        self._B = None
        self.use_linear = True

    def _alloc_tables(self):
        """HBM layout: ONE table of fused rows [R, LD], LD = 2*D floats (a power of two, so a
        row never straddles a 128-byte line): columns 0..D-1 the embedding, column D the FM
        bias-table entry, column D+1 the sparse linear weight, the rest padding.  One line
        fetch per lookup serves all three (tools/bench_embed.py: as separate tables the two
        4-byte gathers cost as much as the row gather).  The reference's per-feature
        variables are strided views of it; `linear_w` is assembled by state_dict().
        (recman_amd/dist.py overrides this with the row-sharded layout.)"""
        spec, dev, Dn, D = self.spec, self.device, self.Dn, self.D
        R = spec.rows
        self.LD = 2 * D
        self.rows = torch.zeros(R, self.LD, dtype=F32, device=dev)
        self.table = self.rows
        self.linear_w_dense = torch.zeros(Dn, dtype=F32, device=dev)
        offs = spec.offsets()
        self.field_off_host = offs
        self._mv = None
        self.field_off = torch.tensor(offs, dtype=I64, device=dev)
        self.lin_off = self.field_off  # sparse one-hot blocks share the table's row numbering
        for name, off, V in zip(spec.sparse_names, offs, spec.feat_sizes):
            self.params[f"{name}_feat_embed"] = self.rows[off: off + V, :D]
            if self.use_bias_tables:
                self.params[f"{name}_feat_bias"] = self.rows[off: off + V, D: D + 1]
        self.params["linear_w_sparse"] = self.rows[:, D + 1]
        self.params["linear_w_dense"] = self.linear_w_dense
        self._set_lin_masks()

    def storage(self):
        """The distinct parameter buffers (for initialisers that fill storage in place)."""
        seen, out = set(), []
        for t in self.params.values():
            base = t if t._base is None else t._base
            if base.data_ptr() not in seen:
                seen.add(base.data_ptr())
                out.append(base)
        return out

    # ------------------------------------------------------------------ storage
    def load_params(self, params):
        """Copies a name -> tensor dict (reference variable names) into the engine."""
        R = self.spec.rows
        for k, v in params.items():
            v = torch.as_tensor(v).to(self.device, F32)
            if k == "linear_w":  # the reference's stacked one-hot blocks / dense columns
                vs, vd = self._lin_from_ref(v.reshape(-1))
                self.params["linear_w_sparse"].copy_(vs)
                self.params["linear_w_dense"].copy_(vd)
                continue
            if k not in self.params:
                raise KeyError(f"unknown variable {k!r}")
            dst = self.params[k]
            dst.copy_(v.reshape(dst.shape))

    def state_dict(self):
        """name -> tensor under the reference's variable names (contiguous copies)."""
        out = {}
        for k, v in self.params.items():
            if k in ("linear_w_sparse", "linear_w_dense"):
                continue
            out[k] = v.detach().clone().contiguous()
        if "linear_w_sparse" in self.params:
            out["linear_w"] = self._lin_to_ref(self.params["linear_w_sparse"].detach().reshape(-1),
                                               self.params["linear_w_dense"].detach().reshape(-1)).view(-1, 1)
        return out

    def _set_lin_masks(self):
        """linear_features subsets: the linear weights of the other features stay at their zero
        initial value (W is zero-initialised, layers.py:318-328) because their gradient is masked."""
        mf, md = self.spec.lin_masks()
        dev = self.device
        self.lin_field_mask = None if mf is None else torch.tensor(mf, dtype=F32, device=dev)
        self.lin_dense_mask = None if (md is None or not md) else torch.tensor(md, dtype=F32, device=dev)

    def _lin_to_ref(self, v_rows, v_dense):
        """([R] in table-row order, [Dn]) -> the reference's linear_w: its one-hot blocks and dense
        columns in its order (default: sparse, value, multi-valued, dense; or `linear_features`)."""
        if self.spec.linear_names is None and not self.spec.scratch_names:
            return torch.cat([v_rows, v_dense])
        return torch.cat([v_rows[p[1]: p[1] + p[2]] if p[0] == "s" else v_dense[p[1]: p[1] + 1]
                          for p in self.spec.lin_ref_layout()])

    def _lin_from_ref(self, v_ref):
        """The inverse: (rows [R], dense [Dn]); entries of features outside `linear_features` are 0."""
        R, Dn = self.spec.rows, self.Dn
        if self.spec.linear_names is None and not self.spec.scratch_names:
            return v_ref[:R], v_ref[R:]
        vs, vd = v_ref.new_zeros(R), v_ref.new_zeros(Dn)
        at = 0
        for p in self.spec.lin_ref_layout():
            if p[0] == "s":
                vs[p[1]: p[1] + p[2]] = v_ref[at: at + p[2]]
                at += p[2]
            else:
                vd[p[1]] = v_ref[at]
                at += 1
        if at != v_ref.numel():
            raise ValueError(f"linear_w has {v_ref.numel()} entries, the linear features need {at}")
        return vs, vd

    def to_reference_names(self, d):
        """Merges the internal linear_w_sparse / linear_w_dense entries into `linear_w`."""
        d = dict(d)
        if "linear_w_sparse" in d:
            d["linear_w"] = self._lin_to_ref(d.pop("linear_w_sparse").reshape(-1),
                                             d.pop("linear_w_dense").reshape(-1)).view(-1, 1)
        return d

    def _alloc(self, B):
        if self._B == B:
            return
        self._B = B
        dev = self.device
        self.E = torch.empty(B, self.F, self.D, dtype=F32, device=dev)
        self.d_rows = torch.empty(B, self.F, self.D, dtype=F32, device=dev)
        self.fm_sum = torch.empty(B, self.D, dtype=F32, device=dev)
        self.fm_logit = torch.empty(B, dtype=F32, device=dev)
        self.lin_logit = torch.empty(B, dtype=F32, device=dev)
        self.logit = torch.empty(B, dtype=F32, device=dev)
        self.pred = torch.empty(B, dtype=F32, device=dev)
        self.dlogit = torch.empty(B, dtype=F32, device=dev)
        self.loss = torch.zeros(1, dtype=F32, device=dev)
        self.loss_part = torch.empty((B + 31) // 32, dtype=F32, device=dev)  # fused head: per-tile loss sums
        self.ws = torch.empty(256 * 1024, dtype=F32, device=dev)
        self.mv_fields = [f for f, n in enumerate(self.spec.sparse_names) if n in self.spec.scratch_names]
        self._alloc_mv(B)
        self._alloc_model(B)

    def _alloc_mv(self, B):
        dev = self.device
        if self.mv_fields:
            # per-batch pooled rows of the multi-valued features live in a scratch block that is
            # addressed AS ROWS OF THE TABLE: its start is aligned so that (scratch - table) is a
            # whole number of rows, and the gather kernel reads row `base + b` like any other
            LD = self.LD
            raw = torch.zeros((len(self.mv_fields) * B + 2) * LD, dtype=F32, device=dev)
            shift = ((self.rows.data_ptr() - raw.data_ptr()) // 4) % LD
            self._mv_raw = raw
            self.mv_scratch = raw[shift: shift + len(self.mv_fields) * B * LD].view(len(self.mv_fields), B, LD)
            base = (self.mv_scratch.data_ptr() - self.rows.data_ptr()) // (4 * LD)
            assert (self.mv_scratch.data_ptr() - self.rows.data_ptr()) % (4 * LD) == 0
            self.field_off_mv = self.field_off.clone()
            for j, f in enumerate(self.mv_fields):
                self.field_off_mv[f] = base + j * B
            self.idx_mv = torch.zeros(B, self.F, dtype=I64, device=dev)
            self._arange = torch.arange(B, dtype=I64, device=dev)

    def _alloc_model(self, B):
        pass

    # ------------------------------------------------------------------ forward
    def _embed(self, idx, dense, want_fm, masks, lin_w=None):
        m = masks or {}
        fm_masks = m.get("fm", (None, None))
        D = self.D
        foff = self.field_off
        if self.mv_fields:
            mv = self._mv
            if mv is None:
                raise ValueError(f"features {self.spec.scratch_names} need their ids / values (mv=...)")
            self.idx_mv.copy_(idx)
            for j, f in enumerate(self.mv_fields):
                offsets, ids, vals = self._mv_entry(f)
                ops.pool_rows(self.rows, int(self.field_off_host[f]), D, offsets, ids, self.mv_scratch[j],
                              vals=vals)
                self.idx_mv[:, f] = self._arange
            idx, foff = self.idx_mv, self.field_off_mv
        ops.embed_fwd(
            idx, self.rows, foff, D=D, table_ld=self.LD,
            bias_col=D if (want_fm and self.use_bias_tables) else None,
            lin_col=D + 1 if self.use_linear else None,
            lin_w_dense=self.linear_w_dense if (self.use_linear and self.Dn) else None,
            dense=dense if (self.use_linear and self.Dn) else None,
            lin_w0=self.params["linear_w0"] if self.use_linear else None,
            mask_b=fm_masks[0] if want_fm else None, mask_e=fm_masks[1] if want_fm else None,
            E=self.E, fm_sum=self.fm_sum if want_fm else None,
            fm_logit=self.fm_logit if want_fm else None,
            lin_logit=self.lin_logit if self.use_linear else None,
            # hp["table_row_reuse"] = "stream" (default): ids touch a row about once per batch
            # (hashed ids over a table far beyond the caches) -> non-temporal row loads keep E in the
            # caches for the MLP / CIN kernels; "cache": heavy id reuse (Zipf-like), plain loads
            stream_rows=self.hp.get("table_row_reuse", "stream") == "stream")

    def _mv_entry(self, f):
        """(offsets, ids, vals) of scratch-row field f from the mv dict: a multi-valued feature
        gives (offsets, ids); a value feature (offsets, ids, vals) with one id per example."""
        name = self.spec.sparse_names[f]
        ent = self._mv[name]
        vals = ent[2] if len(ent) > 2 else None
        if (name in self.spec.value_names) != (vals is not None):
            raise ValueError(f"feature {name}: value features take (offsets, ids, vals), "
                             "multi-valued ones (offsets, ids)")
        return ent[0], ent[1], vals

    def forward(self, idx, dense=None, training=False, masks=None, manual_weights=None, mv=None):
        """-> (logit [B], pred [B]).  training=False disables dropout and (as the
        reference does, layers.py:338-345) adds the per-feature manual weights (a vector in
        the reference's linear_w order).  mv: name -> (offsets, ids) device tensors of the
        multi-valued features."""
        self._alloc(idx.shape[0])
        self._mv = mv
        backup = None
        if manual_weights is not None:
            # W + weights for this call only: exact restore from a copy (an add/subtract pair
            # would drift by an ulp per predict call)
            mw = manual_weights.to(self.device, F32).reshape(-1)
            R = self.spec.rows
            backup = (self.params["linear_w_sparse"].clone(), self.linear_w_dense.clone())
            ms, md = self._lin_from_ref(mw)
            self.params["linear_w_sparse"].add_(ms)
            self.linear_w_dense.add_(md)
        try:
            branches = self._branches_fwd(idx, dense, training, masks, None)
            ops.logit_loss(branches, task=self.task, logit=self.logit, pred=self.pred)
        finally:
            if backup is not None:
                self.params["linear_w_sparse"].copy_(backup[0])
                self.linear_w_dense.copy_(backup[1])
        return self.logit, self.pred

    # --------------------------------------------------------- forward+backward
    def fwd_bwd(self, idx, dense, y, masks=None, mv=None):
        """One training step's forward + backward.  Returns the loss tensor [1]
        (data loss + l2 terms).  Gradients: self.grads (dense parameters),
        self.d_rows [B,F,D] + idx (embedding rows, IndexedSlices form), self.dlogit
        (per-occurrence gradient of the bias-table / sparse linear entries)."""
        B = idx.shape[0]
        self._alloc(B)
        self._mv = mv
        yk = dict(y=y) if y.dtype == I64 else dict(y_f=y)
        scale = getattr(self, "grad_scale", 1.0)
        # models whose DNN is the last branch of the forward offer it the fused head (rm_mlp_tail):
        # final logit, prediction, loss and dLoss/dlogit in the MLP kernel's epilogue
        self._head_req = dict(task=self.task, grad_scale=scale, logit=self.logit, pred=self.pred,
                              dlogit=self.dlogit, loss_partial=self.loss_part, loss=self.loss, **yk)
        self._head_done = False
        try:
            branches = self._branches_fwd(idx, dense, True, masks, None)
        finally:
            self._head_req = None
        if not self._head_done:
            ops.logit_loss(branches, task=self.task, logit=self.logit, pred=self.pred,
                           dlogit=self.dlogit, loss=self.loss, workspace=self.ws, **yk)
            if scale != 1.0:
                # this batch is one of several micro-batches of a step: its gradients are its share
                # of the full-batch mean (recman_amd/dist.py)
                self.dlogit.mul_(scale)
        self._lin_done = False
        self._branches_bwd(idx, dense, self.dlogit, masks)
        if self.use_linear and not self._lin_done:
            ops.linear_dense_bwd(self.dlogit, dense if self.Dn else None,
                                 self.grads["linear_w_dense"] if self.Dn else None,
                                 self.grads["linear_w0"], self.ws)
        if self.use_linear and self.Dn and self.lin_dense_mask is not None:
            self.grads["linear_w_dense"].mul_(self.lin_dense_mask)  # linear_features subset
        return self._add_l2(self.loss)

    fuse_head = True  # tests switch it off to compare against rm_logit_loss + the chain kernel

    def _mlp_last(self, mlp, xe, xd, keep, masks, others):
        """The DNN as the LAST branch of the forward: `others` = the (logit, coefficient) pairs
        already computed.  Hands the fused head to the MLP when fwd_bwd asked for it and at most two
        other branches exist; returns the DNN logit."""
        req = getattr(self, "_head_req", None) if self.fuse_head else None
        head = dict(req, branches=others, coef_mlp=1.0) if (req is not None and len(others) <= 2) else None
        out = mlp.forward(xe, xd, keep, masks, head=head)
        self._head_done = bool(head is not None and mlp.head_done)
        return out

    def _lin_grads(self):
        """(d linear_w_dense, d linear_w0) for MLP.backward to fill when the linear term is on."""
        if not (self.use_linear and self.Dn):
            return None
        return (self.grads["linear_w_dense"], self.grads["linear_w0"])

    def _add_l2(self, loss):
        hp = self.hp
        total = loss
        if hp.get("lazy_l2", False):
            # LAZY l2 (the row-wise optimizer adds reg * row for the rows a batch touches): the
            # table's l2 terms are neither summed into the loss (a pass over the whole table per step) nor turned
            # into a dense gradient; the linear term's DENSE weights keep their exact l2 gradient
            reg = hp.get("linear_l2_reg", 0.0)
            if reg and self.use_linear and self.Dn:
                self.grads["linear_w_dense"].add_(self.linear_w_dense, alpha=reg)
                total = total + reg * 0.5 * self.linear_w_dense.square().sum()
            return self._add_l2_model(total)
        reg = hp.get("embedding_l2_reg", 0.0)
        if reg:
            total = total + reg * 0.5 * self.rows[:, : self.D].square().sum()
        reg = hp.get("linear_l2_reg", 0.0)
        if reg and self.use_linear:
            total = total + reg * 0.5 * (self.params["linear_w_sparse"].square().sum()
                                         + self.linear_w_dense.square().sum())
        return self._add_l2_model(total)

    def _add_l2_model(self, total):
        return total

    # ------------------------------------------------------------ measurement
    def roofline_probes(self, idx, dense, y):
        """The hand-written hot kernels of this model as [{name, symbol, fn, work, bound}], the
        dominant one first: `fn` launches the kernel once on the current stream, `work` is its
        algorithmic bytes (bound "hbm") or flops (bound "mfma") per launch, `symbol`
        the kernel's name in a rocprofv3 trace."""
        self._alloc(idx.shape[0])
        fm = self._has_fm()
        return [dict(name="embed_fwd_fused_kernel (rm_embed_fwd: gather + FM + linear)",
                     symbol="embed_fwd_fused_kernel", fn=lambda: self._embed(idx, dense, fm, None),
                     work=self._embed_fwd_bytes(idx.shape[0], fm), bound="hbm")]

    def roofline_probe_all(self, idx, dense, y, iters=20):
        """Times every roofline_probes() kernel with HIP events on the stream it is launched on and
        prices it against its roofline."""
        out = []
        for p in self.roofline_probes(idx, dense, y):
            fn = p["fn"]
            # warm-up by TIME, not by count: the chip raises its clock only after ~10 ms of sustained load
            # (in-kernel s_memtime / s_memrealtime: a 0.37 ms GEMM launch runs at 1.97 GHz after 15
            # back-to-back launches and at 2.35 GHz after 30, profiles/r02_dense_gemm.md) - the training
            # loop the kernel belongs to runs sustained, so it is priced at the sustained clock
            t_warm = torch.cuda.Event(enable_timing=True)
            t_warm.record()
            for _ in range(3):
                fn()
            while True:
                for _ in range(5):
                    fn()
                t_now = torch.cuda.Event(enable_timing=True)
                t_now.record()
                t_now.synchronize()
                if t_warm.elapsed_time(t_now) >= 40.0:
                    break
            iters_p = max(iters, 50) if p["bound"] == "hbm" else iters  # (the MFMA kernels take milliseconds each)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for _ in range(iters_p)]
            for a, b in ev:
                a.record()
                fn()
                b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in ev)
            ms = sum(ts) / len(ts)
            if p["bound"] == "hbm":
                achieved, peak, unit = p["work"] / (ms * 1e-3) / 1e9, 8000.0, "GB/s"
            else:
                # (peak: the dense f32 MFMA rate, or the probe's own - the bf16 pipe for the split-operand GEMM)
                achieved, peak, unit = p["work"] / (ms * 1e-3) / 1e12, p.get("peak", 157.3), "TFLOP/s"
            rec = {"kernel": p["name"], "symbol": p["symbol"], "bound": p["bound"],
                   "achieved": round(achieved, 2), "peak": peak, "unit": unit,
                   "frac": round(achieved / peak, 4), "traffic": None,
                   "avg_launch_us": round(ms * 1e3, 2), "min_launch_us": round(ts[0] * 1e3, 2),
                   "median_launch_us": round(ts[len(ts) // 2] * 1e3, 2), "algorithmic_per_launch": p["work"],
                   "timing": f"hipEvent pairs, {iters_p} launches (mean; min and median beside it)"}
            if "extra" in p:
                rec.update(p["extra"](ms))
            if "work_min" in p:
                # the same launch priced on the bytes the fused kernel itself has to move
                rec["algorithmic_min_per_launch"] = p["work_min"]
                rec["frac_min"] = round(p["work_min"] / (ms * 1e-3) / 1e9 / 8000.0, 4)
            out.append(rec)
        return out

    def roofline_probe(self, idx, dense, y, iters=20):
        """The dominant kernel's roofline record (roofline_probe_all()[0])."""
        return self.roofline_probe_all(idx, dense, y, iters)[0]

    def optimizer_probe(self, idx, iters=10, more_ids=()):
        """Times the separately-reported optimizer step on the gradients of the last fwd_bwd: the
        row-wise step on the touched table rows + the dense parameters in one launch; checks that two
        identical steps from the same state give bit-identical tables.  more_ids: further id batches -
        the timed steps then cycle through all of them (every step touches other rows, as in fit())."""
        from .optim import FusedDenseOptimizer, SparseTableOptimizer

        ids = [idx] + list(more_ids)
        sopt, dopt = SparseTableOptimizer(self, "adam", 1e-3), FusedDenseOptimizer(self, "adam", 1e-3)
        # determinism: the same step twice from the same state (rows + moments restored in between)
        rows0, mom0 = self.rows.clone(), sopt.mom.clone()
        sopt.step(idx)
        rows1, mom1 = self.rows.clone(), sopt.mom.clone()
        self.rows.copy_(rows0)
        sopt.mom.copy_(mom0)
        sopt.t = 0
        sopt.step(idx)
        same = bool(torch.equal(self.rows, rows1) and torch.equal(sopt.mom, mom1))
        del rows0, mom0, rows1, mom1
        iters = max(iters, 2 * len(ids))
        for i in range(3):
            sopt.step(ids[i % len(ids)])
            dopt.step()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        for i in range(iters):
            sopt.step(ids[i % len(ids)])
        ev[1].record()
        for _ in range(iters):
            dopt.step()
        ev[2].record()
        for i in range(iters):
            sopt.prepare(ids[i % len(ids)])
        ev[3].record()
        torch.cuda.synchronize()
        sopt._prepared = None
        ms_s, ms_d = ev[0].elapsed_time(ev[1]) / iters, ev[1].elapsed_time(ev[2]) / iters
        ms_p = ev[2].elapsed_time(ev[3]) / iters
        return {"ms": round(ms_s + ms_d, 4), "sparse_rows_ms": round(ms_s, 4), "dense_params_ms": round(ms_d, 4),
                "sort_ms": round(ms_p, 4), "apply_ms": round(ms_s - ms_p, 4),
                "bit_identical_rerun": same, "id_batches_rotated": len(ids),
                "what": "row-wise lazy Adam on the touched table rows (rm_sparse_optimizer_step: stable sort by "
                        "row, duplicates summed in occurrence order, no float atomics) + Adam on the dense "
                        "parameters in one launch (rm_dense_optimizer_step), timed back to back; sort_ms is the "
                        "id-only part (rm_sparse_optimizer_prepare) that fit() issues on a side stream beside "
                        "fwd+bwd; NOT part of value",
                "roofline": sopt.roofline(idx, ms_s)}

    def _embed_fwd_bytes(self, B, fm):
        F, D, Dn = self.F, self.D, self.Dn
        per = F * 8 + 2 * F * 4 * D  # idx read, rows read, E written
        if fm:
            per += F * 4 + 4 * D + 4  # bias entries, S written, fm_logit written
        if self.use_linear:
            per += F * 4 + Dn * 4 + 4  # linear entries, dense columns, lin_logit
        return B * per

    # ----------------------------------------------- dense views of the sparse grads
    def dense_grads(self, idx, reference_names=False):
        """Densifies the sparse gradients of the last fwd_bwd (scatter-add with float
        atomics) and adds the l2 terms: name -> tensor with the keys / shapes of self.params
        (or of state_dict() with reference_names=True).  What TF's IndexedSlices + dense l2
        gradient add up to (layers.py:188-193)."""
        hp, D = self.hp, self.D
        out = {k: v.clone() for k, v in self.grads.items()}
        R = self.spec.rows
        foff = self.field_off
        if self.mv_fields:
            # occurrences of multi-valued fields go to a dummy row R here; their gradient is
            # scattered to the tag rows below (rm_pool_rows_bwd)
            idx = idx.clone()
            foff = self.field_off.clone()
            for f in self.mv_fields:
                idx[:, f] = 0
                foff[f] = R
        d_table = torch.zeros(R + 1, D, dtype=F32, device=self.device)
        ops.scatter_add_rows(d_table, idx, foff, rows=self.d_rows)
        offs = self.spec.offsets()
        d_bias = None
        g_bias_occ = None
        if self.use_bias_tables:
            d_bias = torch.zeros(R + 1, dtype=F32, device=self.device)
            if self._has_fm():
                if getattr(self, "d_bias", None) is not None:
                    ops.scatter_add_rows(d_bias, idx, foff, rows=self.d_bias, width=1, ld=1)
                    g_bias_occ = self.d_bias
                else:
                    ops.scatter_add_rows(d_bias, idx, foff, g_row=self.dlogit)
        d_lin = torch.zeros(R + 1, dtype=F32, device=self.device)
        if self.use_linear:
            ops.scatter_add_rows(d_lin, idx, foff, g_row=self.dlogit)
        for f in self.mv_fields:
            offsets, ids, vals = self._mv_entry(f)
            gb = None
            if d_bias is not None and self._has_fm():
                gb = g_bias_occ[:, f].contiguous() if g_bias_occ is not None else self.dlogit
            ops.pool_rows_bwd(self.d_rows[:, f, :], gb, self.dlogit if self.use_linear else None, D,
                              offsets, ids, offs[f], d_table, d_bias if gb is not None else None,
                              d_lin if self.use_linear else None, vals=vals)
        d_table, d_lin = d_table[:R], d_lin[:R]
        if self.lin_field_mask is not None:  # linear_features subset
            for f, (off, V) in enumerate(zip(offs, self.spec.feat_sizes)):
                if self.spec.sparse_names[f] not in self.spec.linear_names:
                    d_lin[off: off + V] = 0
        if d_bias is not None:
            d_bias = d_bias[:R]
        reg = hp.get("embedding_l2_reg", 0.0)
        if reg:
            d_table.add_(self.rows[:, :D], alpha=reg)
        for name, off, V in zip(self.spec.sparse_names, offs, self.spec.feat_sizes):
            out[f"{name}_feat_embed"] = d_table[off: off + V]
            if d_bias is not None:
                out[f"{name}_feat_bias"] = d_bias[off: off + V].view(V, 1)
        if self.use_linear:
            reg = hp.get("linear_l2_reg", 0.0)
            if reg:
                d_lin.add_(self.params["linear_w_sparse"], alpha=reg)
                out["linear_w_dense"] = out["linear_w_dense"] + reg * self.linear_w_dense
        else:
            out["linear_w0"] = torch.zeros_like(self.grads["linear_w0"])
            out["linear_w_dense"] = torch.zeros_like(self.grads["linear_w_dense"])
        out["linear_w_sparse"] = d_lin
        self._dense_grads_model(out)
        return self.to_reference_names(out) if reference_names else out

    def _dense_grads_model(self, out):
        pass

    def _has_fm(self):
        return False


# DeepFM's fwd_bwd through rm_deepfm_step (one kernel) where it applies; False = rm_embed_mlp_fwd + rm_mlp_bwd
STEP_FUSION_DEFAULT = True


class DeepFMEngine(Engine):
    """DeepFM._init_graph (DeepFM.py:107-158): final = linear + fm + dnn."""

    model = "deepfm"
    use_bias_tables = True

    def __init__(self, spec, embedding_size, hp, task="classification", device="cuda"):
        super().__init__(spec, embedding_size, hp, task, device)
        self.use_fm = bool(hp.get("use_fm", True))
        self.use_deep = bool(hp.get("use_deep", True))
        assert self.use_fm or self.use_deep  # DeepFM.py:54
        self.mlp = None
        if self.use_deep:
            self.mlp = MLP(self.params, self.grads, self.FD, self.Dn, hp["deep_hidden_units"],
                           hp.get("deep_activation", "relu"), self.device)
            self.mlp.stream_d_rows = hp.get("d_rows_reuse", "cache") == "stream"
            self.mlp.dense_gemm = hp.get("dense_gemm", "bf16x6")

    def _has_fm(self):
        return self.use_fm

    def _front_fused(self, m, lin_w):
        """The one-kernel front (rm_embed_mlp_fwd: gather + FM + linear + MLP + head) covers this call:
        plain id features on the fused-row table, a skinny MLP, no dropout masks."""
        hp = self.hp
        if not (self.use_deep and self.use_linear and hp.get("front_fusion", True)) or self.mv_fields:
            return False
        if lin_w is not None or m.get("dnn") is not None or any(x is not None for x in m.get("fm", (None, None))):
            return False
        if getattr(self, "_front_ok", None) is None:
            self._front_ok = bool(self.mlp.fused_ok and ops.embed_mlp_fwd_supported(
                self.F, self.D, self._front_ld(), self.Dn, self.mlp.hidden))
        return self._front_ok

    def _front_ld(self):
        return self.LD

    def _front_table(self, idx):
        """(ids, rows, field offsets, row stride, non-temporal row loads) the one-kernel front gathers from."""
        return idx, self.rows, self.field_off, self.LD, self.hp.get("table_row_reuse", "stream") == "stream"

    def _front_fwd(self, idx, dense, branches):
        mlp, p = self.mlp, self.params
        B, n = idx.shape[0], len(self.mlp.hidden)
        mlp._alloc(B, idx.device)
        mlp.keep, mlp.masks = [1] * (n + 1), [None] * (n + 1)
        mlp.xe, mlp.xd, mlp.fused = self.E.view(-1, self.FD), (dense if self.Dn else None), True
        req = getattr(self, "_head_req", None) if self.fuse_head else None
        head = dict(req, branches=branches, coef_mlp=1.0) if req is not None else None
        mlp.tail = ops.mlp_tail(B, dh=mlp.dhb, **head) if head is not None else None
        mlp.head_done = head is not None
        pre = mlp.prefix
        ids, rows, foff, ld, stream_rows = self._front_table(idx)
        ops.embed_mlp_fwd(
            ids, rows, foff, self.D, ld, dense if self.Dn else None,
            [p[f"{pre}dnn_layer_{i}_weights"] for i in range(n)], [p[f"{pre}dnn_layer_{i}_bias"] for i in range(n)],
            p[f"{pre}dnn_w"].view(-1), p[f"{pre}dnn_w0"], mlp.act, self.E, mlp.hb, mlp.out.view(B),
            want_bias=self.use_fm and self.use_bias_tables, want_lin=True,
            lin_w_dense=self.linear_w_dense if self.Dn else None, lin_w0=p["linear_w0"],
            fm_sum=self.fm_sum if self.use_fm else None, fm_logit=self.fm_logit if self.use_fm else None,
            lin_logit=self.lin_logit, stream_rows=stream_rows, tail=mlp.tail)
        self._head_done = mlp.head_done
        return mlp.out.view(B)

    # ---- the whole step in one kernel (rm_deepfm_step)
    step_fusable = True  # the row-sharded subclass keeps its own fwd_bwd (recman_amd/dist.py)

    def _step_fused(self, masks, mv):
        """rm_deepfm_step covers this call: everything _front_fused asks for, plus FM + bias tables on, two
        hidden layers, no dropout masks.  hp["step_fusion"] overrides STEP_FUSION_DEFAULT."""
        if not (self.step_fusable and self.hp.get("step_fusion", STEP_FUSION_DEFAULT)) or masks or mv is not None:
            return False
        if not (self.use_fm and self.use_bias_tables and self._front_fused({}, None)):
            return False
        if getattr(self, "_step_ok", None) is None:
            self._step_ok = bool(ops.deepfm_step_supported(self.F, self.D, self.LD, self.Dn, self.mlp.hidden))
        return self._step_ok

    def _step_call(self, idx, dense, y, skip_finish=False):
        mlp, p, g = self.mlp, self.params, self.grads
        if getattr(self, "_step_ws", None) is None:
            self._step_ws = torch.zeros(ops.deepfm_step_workspace(self.F, self.Dn), dtype=F32, device=self.device)
        pre, n = mlp.prefix, len(mlp.hidden)
        ops.deepfm_step(
            idx, self.rows, self.field_off, self.D, self.LD, dense if self.Dn else None, y,
            [p[f"{pre}dnn_layer_{i}_weights"] for i in range(n)], [p[f"{pre}dnn_layer_{i}_bias"] for i in range(n)],
            p[f"{pre}dnn_w"].view(-1), p[f"{pre}dnn_w0"], self.linear_w_dense if self.Dn else None, p["linear_w0"],
            mlp.act, self.task, self.d_rows, self.logit, self.pred, self.dlogit, self.loss,
            [g[f"{pre}dnn_layer_{i}_weights"] for i in range(n)], [g[f"{pre}dnn_layer_{i}_bias"] for i in range(n)],
            g[f"{pre}dnn_w"].view(-1), g[f"{pre}dnn_w0"], g["linear_w_dense"] if self.Dn else None, g["linear_w0"],
            self._step_ws, grad_scale=getattr(self, "grad_scale", 1.0),
            # (plain row loads unless asked otherwise: the reason for streaming them - keeping E in the caches for the
            # MLP kernels - is gone with E; measured 97 vs 102 us per step, uniform ids)
            stream_rows=self.hp.get("step_row_loads", "cache") == "stream",
            stream_d_rows=self.hp.get("d_rows_reuse", "cache") == "stream", skip_finish=skip_finish)

    def fwd_bwd(self, idx, dense, y, masks=None, mv=None):
        B = idx.shape[0]
        self._alloc(B)
        if not self._step_fused(masks, mv):
            return super().fwd_bwd(idx, dense, y, masks=masks, mv=mv)
        self._mv = None
        self._step_call(idx, dense, y)
        self.d_bias = None
        self._head_done = self._lin_done = True
        g = self.grads
        if self.Dn and self.lin_dense_mask is not None:
            g["linear_w_dense"].mul_(self.lin_dense_mask)  # linear_features subset
        reg = self.hp.get("deep_l2_reg", 0.0)
        if reg:
            self.mlp.add_l2_grads(reg)
        return self._add_l2(self.loss)

    def roofline_probes(self, idx, dense, y):
        probes = super().roofline_probes(idx, dense, y)
        self._alloc(idx.shape[0])
        if self._step_fused(None, None):
            # the step's dominant (only large) kernel.  Algorithmic bytes: SURVEY.md 8d's embed+FM forward AND
            # backward figure, 8,952 B per example at F = 26, D = 16 (idx, rows, E, S, g, dE, gradient rows) - the
            # path this kernel replaces end to end; `work_min` = what the fused kernel itself has to move (ids, 72
            # useful bytes per looked-up row, dense inputs, labels in; row gradients, logit / pred / dlogit out)
            B, F, D, Dn = idx.shape[0], self.F, self.D, self.Dn
            fwd = B * F * (8 + 4 * D + 4) + B * 4 + B * F * 4 * D
            bwd = B * F * (8 + 4 * D + 4 * D + 4 * D + 4) + B * 4
            work_min = B * (F * (8 + 4 * D + 8) + 4 * Dn + y.element_size() + F * 4 * D + 3 * 4)
            step = [dict(name="deepfm_step_kernel (rm_deepfm_step: gather + FM + linear + MLP + loss + every gradient)",
                         symbol="deepfm_step_kernel", fn=lambda: self._step_call(idx, dense, y, skip_finish=True),
                         work=fwd + bwd, work_min=work_min, bound="hbm")]
            return step + probes
        if not self._front_fused({}, None):
            return probes
        # the step's dominant kernel is the one-kernel front; its algorithmic bytes = rm_embed_fwd's
        #  + what the MLP and the head move per example on top of x: the dense columns when
        # the linear term has not already counted them, h_l written, dh_l written, the label, logit /
        # pred / dlogit written (E is NOT read back: that is the point of the fusion)
        B, n = idx.shape[0], len(self.mlp.hidden)
        yk = dict(y=y) if y.dtype == I64 else dict(y_f=y)

        def run():
            self._head_req = dict(task=self.task, grad_scale=1.0, logit=self.logit, pred=self.pred,
                                  dlogit=self.dlogit, loss_partial=self.loss_part, loss=self.loss, **yk)
            try:
                branches = [(self.lin_logit, 1.0)] + ([(self.fm_logit, 1.0)] if self.use_fm else [])
                self._front_fwd(idx, dense, branches)
            finally:
                self._head_req = None

        work = self._embed_fwd_bytes(B, self._has_fm()) + B * (2 * n * 32 * 4 + y.element_size() + 4 * 4)
        return [dict(name="embed_mlp_fwd_kernel (rm_embed_mlp_fwd: gather + FM + linear + MLP + head)",
                     symbol="embed_mlp_fwd_kernel", fn=run, work=work, bound="hbm")] + probes

    def _branches_fwd(self, idx, dense, training, masks, lin_w):
        hp = self.hp
        m = masks or {}
        if not training:
            m = {}
        if self._front_fused(m, lin_w):
            branches = [(self.lin_logit, 1.0)]
            if self.use_fm:
                branches.append((self.fm_logit, 1.0))
            self.dnn_logit = self._front_fwd(idx, dense, list(branches))
            branches.append((self.dnn_logit, 1.0))
            return branches
        self._embed(idx, dense, self.use_fm, m, lin_w)
        branches = [(self.lin_logit, 1.0)]
        if self.use_fm:
            branches.append((self.fm_logit, 1.0))
        if self.use_deep:
            n = len(hp["deep_hidden_units"])
            keep = list(hp.get("deep_dropout", [1] * (n + 1))) if training else [1] * (n + 1)
            self.dnn_logit = self._mlp_last(self.mlp, self.E.view(-1, self.FD), dense if self.Dn else None,
                                            keep, m.get("dnn"), list(branches))
            branches.append((self.dnn_logit, 1.0))
        return branches

    def _branches_bwd(self, idx, dense, g, masks):
        m = masks or {}
        fm_masks = m.get("fm", (None, None))
        dE_up = None
        fm_done = False
        self.d_bias = None
        if self.use_deep:
            # fused path: the FM gradient g*(S - E) rides along in the MLP backward
            fuse_fm = self.use_fm and fm_masks[0] is None and fm_masks[1] is None
            fm_done = self.mlp.backward(g, self.d_rows.view(-1, self.FD),
                                        fm_sum=self.fm_sum if fuse_fm else None,
                                        lin_grads=self._lin_grads())
            self._lin_done = self.mlp.lin_done
            dE_up = self.d_rows
        if self.use_fm and fm_masks[0] is not None:
            self.d_bias = torch.empty(idx.shape[0], self.F, dtype=F32, device=self.device)
        if self.use_fm and not fm_done:
            ops.embed_bwd(self.d_rows, E=self.E, fm_sum=self.fm_sum, dE_up=dE_up, g_fm=g,
                          mask_b=fm_masks[0], mask_e=fm_masks[1], d_bias=self.d_bias)
        if self.use_deep:
            reg = self.hp.get("deep_l2_reg", 0.0)
            if reg:
                self.mlp.add_l2_grads(reg)

    def _add_l2_model(self, total):
        reg = self.hp.get("deep_l2_reg", 0.0)
        if reg and self.use_deep:
            total = total + self.mlp.l2(reg)
        return total


class DCNEngine(Engine):
    """DCN._init_graph (DCN.py:99-144): dnn_input feeds the DNN and the CrossNet;
    final = dnn + cross (+ dnn again under strict_reference, DCN.py:140-142)
    (+ linear if use_linear).  CrossNet is absent from the reference (DCN.py:7):
    cross_type "vector" = DCN-v1, x_{l+1} = x0 (x_l . w_l) + b_l + x_l, all layers fused in
    csrc/cross.hip; cross_type "matrix" = x_{l+1} = x0 o (W_l x_l + b_l) + x_l, one f32-MFMA
    GEMM per layer with the cross update as its epilogue (csrc/gemm.hip, RM_DENSE_CROSS)."""

    model = "dcn"
    use_bias_tables = False

    def __init__(self, spec, embedding_size, hp, task="classification", device="cuda"):
        super().__init__(spec, embedding_size, hp, task, device)
        self.use_linear = bool(hp.get("use_linear", True))
        self.L = int(hp.get("cross_layer_num", 3))
        self.dnn_coef = 2.0 if hp.get("strict_reference", False) else 1.0
        d = self.FD + self.Dn
        dev = self.device
        self.mlp = MLP(self.params, self.grads, self.FD, self.Dn, hp["deep_hidden_units"],
                       hp.get("deep_activation", "relu"), dev)
        self.mlp.stream_d_rows = hp.get("d_rows_reuse", "cache") == "stream"
        self.mlp.dense_gemm = hp.get("dense_gemm", "bf16x6")
        self.cross_type = hp.get("cross_type", "vector")
        if self.cross_type not in ("vector", "matrix"):
            raise ValueError(f"cross_type {self.cross_type!r}: 'vector' or 'matrix'")
        self.matrix = self.cross_type == "matrix"
        wshape = (self.L, d, d) if self.matrix else (self.L, d)
        for nm, shape in (("cross_w", wshape), ("cross_b", (self.L, d)), ("cross_w_out", (d, 1))):
            self.params[nm] = torch.zeros(shape, dtype=F32, device=dev)
            self.grads[nm] = torch.zeros(shape, dtype=F32, device=dev)

    def _alloc_model(self, B):
        dev = self.device
        L = self.L
        self.cross_logit = torch.empty(B, dtype=F32, device=dev)
        self.cross_p = torch.empty(B, ops.cross_p_ld(L), dtype=F32, device=dev)  # x0.w_l, x0.w_out
        self.coef = torch.empty(B, 2 * L + 2, dtype=F32, device=dev)
        self._coef_sum = torch.empty(2 * L + 2, dtype=F32, device=dev)
        self._ones_b = torch.ones(B, dtype=F32, device=dev)
        self._cross_wws = torch.empty(max(1, ops.dense_wgrad_workspace(self.FD + self.Dn, L + 1, B)),
                                      dtype=F32, device=dev)
        self.P = torch.empty(self.FD + self.Dn, L + 1, dtype=F32, device=dev)
        self.dxe_dnn = torch.empty(B, self.FD, dtype=F32, device=dev)
        if self.matrix:
            # activations of the matrix cross: rows padded to a multiple of 4 floats (16-byte rows
            # for the GEMM loaders); the pad columns stay zero
            d = self.FD + self.Dn
            dp = (d + 3) // 4 * 4
            z = lambda: torch.zeros(B, dp, dtype=F32, device=dev)  # noqa: E731
            self.cx = [z() for _ in range(L + 1)]   # x_0 .. x_L
            self.cu = [z() for _ in range(L)]       # u_l = W_l x_l + b_l
            self.cg = [z(), z()]                    # dLoss/dx_l, ping-pong
            self.cdu, self.cdx0 = z(), z()
            self.w_out_p = torch.zeros(dp, dtype=F32, device=dev)
            self._dcol = torch.empty(dp, dtype=F32, device=dev)
            self._cfws = torch.empty(ops.dense_filter_workspace(d, d), dtype=F32, device=dev)
            self._cwws = torch.empty(max(1, ops.dense_wgrad_workspace(d, d, B)), dtype=F32, device=dev)

    # ---- matrix cross (DCN-v2 form): one GEMM per layer, cross update in the epilogue ----
    def _cross_matrix_fwd(self, xe, xd):
        p, d = self.params, self.FD + self.Dn
        x0 = self.cx[0]
        x0[:, : self.FD].copy_(xe)
        if self.Dn:
            x0[:, self.FD: d].copy_(xd)
        for l in range(self.L):
            ops.dense_fwd(self.cx[l][:, :d], None, p["cross_w"][l], self.cx[l + 1][:, :d], self._cfws,
                          transposed=True, bias=p["cross_b"][l], epilogue=ops.DENSE_CROSS,
                          aux1=x0[:, :d], aux2=self.cx[l][:, :d], out2=self.cu[l][:, :d])
        self.w_out_p[:d].copy_(p["cross_w_out"].view(-1))
        ops.rowdot(self.cx[self.L], self.w_out_p, None, self.cross_logit)

    def _cross_matrix_bwd(self, g):
        p, gr, d, L = self.params, self.grads, self.FD + self.Dn, self.L
        x0 = self.cx[0]
        ops.linear_dense_bwd(g, self.cx[L], self._dcol, None, self.ws)  # d_w_out = x_L^T g
        gr["cross_w_out"].view(-1).copy_(self._dcol[:d])
        gc, gn = self.cg
        torch.mul(g.view(-1, 1), self.w_out_p.view(1, -1), out=gc)  # dLoss/dx_L
        self.cdx0.zero_()
        for l in range(L - 1, -1, -1):
            torch.mul(gc, x0, out=self.cdu)                 # dLoss/du_l = g_{l+1} o x0
            self.cdx0.addcmul_(gc, self.cu[l])              # dLoss/dx0 += g_{l+1} o u_l
            ops.dense_wgrad(self.cdu[:, :d], None, self.cx[l][:, :d], gr["cross_w"][l], self._cwws)
            ops.linear_dense_bwd(self._ones_b, self.cdu, self._dcol, None, self.ws)
            gr["cross_b"][l].copy_(self._dcol[:d])
            # dLoss/dx_l = du_l W_l + g_{l+1}
            ops.dense_fwd(self.cdu[:, :d], None, p["cross_w"][l], gn[:, :d], self._cfws,
                          epilogue=ops.DENSE_ADD, aux1=gc[:, :d])
            gc, gn = gn, gc
        self.cdx0.add_(gc)
        torch.add(self.cdx0[:, : self.FD], self.dxe_dnn, out=self.d_rows.view(-1, self.FD))

    def _branches_fwd(self, idx, dense, training, masks, lin_w):
        hp = self.hp
        m = (masks or {}) if training else {}
        self._embed(idx, dense, False, m, lin_w)
        xe, xd = self.E.view(-1, self.FD), (dense if self.Dn else None)
        n = len(hp["deep_hidden_units"])
        keep = list(hp.get("deep_dropout", [1] * (n + 1))) if training else [1] * (n + 1)
        self.dnn_logit = self.mlp.forward(xe, xd, keep, m.get("dnn"))
        p = self.params
        if self.matrix:
            self._cross_matrix_fwd(xe, xd)
        else:
            ops.cross_fwd(xe, xd, p["cross_w"], p["cross_b"], p["cross_w_out"].view(-1),
                          self.cross_logit, self.cross_p)
        branches = [(self.dnn_logit, self.dnn_coef), (self.cross_logit, 1.0)]
        if self.use_linear:
            branches.append((self.lin_logit, 1.0))
        return branches

    def _branches_bwd(self, idx, dense, g, masks):
        p, gr = self.params, self.grads
        xe, xd = self.E.view(-1, self.FD), (dense if self.Dn else None)
        g_dnn = g if self.dnn_coef == 1.0 else g * self.dnn_coef
        L = self.L
        # the first dense layer's dW = x0^T dA0 and the cross net's P = x0^T coef read the same x0: one pass
        # (rm_dense_wgrad6 with a second piece of gradient columns) once both operands exist
        fold = (not self.matrix and L + 1 <= 16 and self.mlp.can_defer_wgrad0()
                and self.hp.get("dcn_fold_cross_wgrad", True))
        self.mlp.backward(g_dnn, self.dxe_dnn, defer_wgrad0=fold)
        if self.matrix:
            self._cross_matrix_bwd(g)
            self._cross_l2_grads()
            return
        # cross backward adds the DNN's dx and writes straight into the row-gradient
        # buffer: with no FM term d_rows IS dLoss/dE (no separate embed_bwd launch)
        ops.cross_bwd(p["cross_w"], p["cross_b"], p["cross_w_out"].view(-1), g,
                      self.cross_p, self.d_rows.view(-1, self.FD), self.coef, dx_in_e=self.dxe_dnn)
        # P = x0^T coef[:, :L+1]: a batch-reduction GEMM with x0 = [xe | xd] read in place, and the
        # column sums of coef in one pass (rm_linear_dense_bwd with unit weights)
        if fold:
            self.mlp.wgrad0(G2=self.coef[:, : L + 1], dW2=self.P)
        else:
            ops.dense_wgrad(xe, xd if self.Dn else None, self.coef[:, : L + 1], self.P, self._cross_wws)
        ops.linear_dense_bwd(self._ones_b, self.coef, self._coef_sum, None, self.ws)
        colsum = self._coef_sum[L + 1:]
        ops.cross_param_grads(self.P, colsum, p["cross_w"], p["cross_b"],
                              p["cross_w_out"].view(-1), gr["cross_w"], gr["cross_b"],
                              gr["cross_w_out"].view(-1))
        self._cross_l2_grads()

    def roofline_probes(self, idx, dense, y):
        # the MLP GEMMs dominate the DCN step: layer 0 of the wide DNN, x = [xe | xd] -> H0 (bias +
        # activation fused), on the f32 MFMA roofline; the fused cross kernels on the HBM roofline
        if self.mlp.fused_ok:
            return super().roofline_probes(idx, dense, y)
        B = idx.shape[0]
        self._alloc(B)
        if getattr(self, "_probe_ready", None) != B:
            self.fwd_bwd(idx, dense, y)  # fills E, the layer scalars s, dlogit, dxe_dnn
            self._probe_ready = B
        xe, xd = self.E.view(-1, self.FD), (dense if self.Dn else None)
        W, b = self.params["dnn_layer_0_weights"], self.params["dnn_layer_0_bias"]
        m, p, L, d = self.mlp, self.params, self.L, self.FD + self.Dn
        K, N = W.shape
        m._alloc_dense(xe.device)
        if m._fws6 is not None and ops.dense_fwd6_supported(xe, xd):
            # the product path: fp32 operands split into three bf16 pieces, SIX bf16 MFMAs per k-step (csrc/gemm6.hip)
            # - priced on the bf16 pipe with the six products counted; the fp32 GEMM it stands for beside it
            flops = 2.0 * B * K * N
            probes = [dict(name=f"dense_nn6_kernel (rm_dense_fwd6, DNN layer 0: [{B},{K}] x [{K},{N}] + bias + {m.act}; "
                                "fp32 operands as 3 bf16 pieces, 6 piece products per k-step on the bf16 matrix pipe)",
                           symbol="dense_nn6_kernel",
                           fn=lambda: ops.dense_fwd(xe, xd, W, m.a[0], m._fws, bias=b, act=m.act, ws6=m._fws6),
                           work=6.0 * flops, bound="mfma", peak=2500.0,
                           extra=lambda ms: {"fp32_gemm_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
                                             "fp32_gemm_vs_f32_mfma_peak_157": round(flops / (ms * 1e-3) / 1e12 / 157.3, 4),
                                             "note": "peak = dense bf16 MFMA (2.5 PFLOP/s); achieved counts the six "
                                                     "bf16 products per fp32 product; fp32_gemm_tflops = 2 M K N / time"})]
        else:
            probes = [dict(name=f"dense_nn_kernel (rm_dense_fwd, DNN layer 0: [{B},{K}] x [{K},{N}] + bias + {m.act})",
                           symbol="dense_nn_kernel",
                           fn=lambda: ops.dense_fwd(xe, xd, W, m.a[0], m._fws, bias=b, act=m.act),
                           work=2.0 * B * K * N, bound="mfma")]
        if not self.matrix:
            probes.append(dict(
                name=f"cross_fwd_kernel (rm_cross_fwd, {L} layers fused: x0 [{B},{d}] read once -> logit, p)",
                symbol="cross_fwd_kernel",
                fn=lambda: ops.cross_fwd(xe, xd, p["cross_w"], p["cross_b"], p["cross_w_out"].view(-1),
                                         self.cross_logit, self.cross_p),
                work=B * (d * 4 + 4 + 4 * (L + 1)), bound="hbm"))
            probes.append(dict(
                name="cross_bwd_kernel (rm_cross_bwd: the DNN's dx [B,FD] + the p row read, dx0 written once)",
                symbol="cross_bwd_kernel",
                fn=lambda: ops.cross_bwd(p["cross_w"], p["cross_b"], p["cross_w_out"].view(-1),
                                         self.dlogit, self.cross_p, self.d_rows.view(-1, self.FD),
                                         self.coef, dx_in_e=self.dxe_dnn),
                work=B * (2 * self.FD * 4 + 4 * (2 * L + 2) + 4 * (L + 1) + 4), bound="hbm"))
        return probes

    def _cross_l2_grads(self):
        p, gr = self.params, self.grads
        reg = self.hp.get("deep_l2_reg", 0.0)
        if reg:
            self.mlp.add_l2_grads(reg)
        reg = self.hp.get("cross_layer_l2_reg", 0.0)
        if reg:
            gr["cross_w"].add_(p["cross_w"], alpha=reg)
            gr["cross_w_out"].add_(p["cross_w_out"], alpha=reg)

    def _add_l2_model(self, total):
        reg = self.hp.get("deep_l2_reg", 0.0)
        if reg:
            total = total + self.mlp.l2(reg)
        reg = self.hp.get("cross_layer_l2_reg", 0.0)
        if reg:
            total = total + reg * 0.5 * (self.params["cross_w"].square().sum()
                                         + self.params["cross_w_out"].square().sum())
        return total


class XDeepFMEngine(Engine):
    """xDeepFM._out (xDeepFM.py:47-104): embeddings without bias tables,
    final = linear + cin + dnn.  CIN (layers.py:697-760) runs on the f32 MFMA, one
    kernel per layer forward, three MFMA passes per layer backward (csrc/cin.hip)."""

    model = "xdeepfm"
    use_bias_tables = False

    def __init__(self, spec, embedding_size, hp, task="classification", device="cuda"):
        super().__init__(spec, embedding_size, hp, task, device)
        dev = self.device
        self.units = [int(u) for u in hp["cin_cross_layer_units"]]
        assert len(self.units) > 0  # layers.py:656
        self.cin_act = act_name(hp.get("cin_activation", "leaky_relu"))
        keep = hp.get("cin_dropout")
        if keep is not None and any(k < 1 for k in keep):
            assert len(keep) == len(self.units) + 1  # layers.py:657 (checked only when it matters)
        self.mlp = MLP(self.params, self.grads, self.FD, self.Dn, hp["deep_hidden_units"],
                       hp.get("deep_activation", "leaky_relu"), dev)
        self.mlp.stream_d_rows = hp.get("d_rows_reuse", "cache") == "stream"
        self.mlp.dense_gemm = hp.get("dense_gemm", "bf16x6")
        m = self.F
        self.Hs, self.pool_from, self.pool_col0 = [m], [], []
        final = 0
        for i, size in enumerate(self.units):
            last = i == len(self.units) - 1
            if not last and size % 2:
                raise ValueError("CIN layer sizes before the last must be even (split in halves, layers.py:742-746)")
            H = self.Hs[-1]
            for nm, shape in ((f"cin_filter_{i}", (1, m * H, size)), (f"cin_bias_{i}", (size,))):
                self.params[nm] = torch.zeros(shape, dtype=F32, device=dev)
                self.grads[nm] = torch.zeros(shape, dtype=F32, device=dev)
            self.pool_from.append(0 if last else size // 2)
            self.pool_col0.append(final)
            final += size if last else size // 2
            self.Hs.append(size // 2)
        self.P = final
        for nm, shape in (("cin_w", (final, 1)), ("cin_w0", (1,))):
            self.params[nm] = torch.zeros(shape, dtype=F32, device=dev)
            self.grads[nm] = torch.zeros(shape, dtype=F32, device=dev)

    def _alloc_model(self, B):
        dev, m, D = self.device, self.F, self.D
        self.maps = [torch.empty(B, n, D, dtype=F32, device=dev) for n in self.units]
        self.dxk = [None] + [torch.empty(B, self.Hs[i], D, dtype=F32, device=dev)
                             for i in range(1, len(self.units))]
        self.pooled = torch.empty(B, self.P, dtype=F32, device=dev)
        self.cin_logit = torch.empty(B, dtype=F32, device=dev)
        fw = max(ops.cin_filter_workspace(m, self.Hs[i], n) for i, n in enumerate(self.units))
        bw = max(ops.cin_bwd_workspace(B, m, self.Hs[i], n, D) for i, n in enumerate(self.units))
        self.cin_fws = torch.empty(fw, dtype=F32, device=dev)
        self.cin_bws = torch.empty(bw, dtype=F32, device=dev)
        # the layers rm_cin_layer_fwd6 covers run on the bf16 matrix pipe with split fp32 operands (csrc/cin6.hip)
        # unless cin_gemm = "f32"
        self.cin_fws6 = None
        if self.hp.get("cin_gemm", "bf16x6") == "bf16x6":
            f6 = max(ops.cin_filter_workspace6(m, self.Hs[i], n, D) for i, n in enumerate(self.units))
            if f6 > 0:
                self.cin_fws6 = torch.empty(f6, dtype=F32, device=dev)

    def _cin_fwd(self, keep=None, masks=None):
        """CIN.__call__ (layers.py:697-760).  keep / masks: the L+1 keep probabilities and 0/1
        masks of cin_dropout (input E, then every layer's maps, layers.py:708,740); a dropped
        layer is scaled in place after its kernel and its pooled columns are re-summed."""
        p = self.params
        L = len(self.units)
        on = [bool(keep is not None and masks is not None and keep[i] < 1 and masks[i] is not None)
              for i in range(L + 1)]
        self._cin_drop = (keep, masks, on) if any(on) else None
        X0 = self.E
        if on[0]:
            self.E_cin = self.E * (masks[0] / keep[0])
            X0 = self.E_cin
        self._cin_x0 = X0
        xk = X0
        for i, n in enumerate(self.units):
            ops.cin_layer_fwd(X0, xk, self.Hs[i], p[f"cin_filter_{i}"][0], p[f"cin_bias_{i}"],
                              self.cin_act, self.maps[i], self.cin_fws,
                              pooled=None if on[i + 1] else self.pooled,
                              pool_col0=self.pool_col0[i], pool_from=self.pool_from[i], ws6=self.cin_fws6,
                              first6=self.hp.get("cin_first_layer", "bf16x6") == "bf16x6")
            if on[i + 1]:
                pf, c0 = self.pool_from[i], self.pool_col0[i]
                self.maps[i].mul_(masks[i + 1] / keep[i + 1])
                torch.sum(self.maps[i][:, pf:, :], dim=2, out=self.pooled[:, c0: c0 + n - pf])
            xk = self.maps[i]
        ops.rowdot(self.pooled, p["cin_w"].view(-1), p["cin_w0"], self.cin_logit)

    def _branches_fwd(self, idx, dense, training, masks, lin_w):
        hp = self.hp
        m = (masks or {}) if training else {}
        self._embed(idx, dense, False, m, lin_w)
        ck = list(hp.get("cin_dropout") or []) if training else []
        self._cin_fwd(ck if ck and any(k < 1 for k in ck) else None, m.get("cin"))
        n = len(hp["deep_hidden_units"])
        keep = list(hp.get("deep_dropout", [1] * (n + 1))) if training else [1] * (n + 1)
        self.dnn_logit = self._mlp_last(self.mlp, self.E.view(-1, self.FD), dense if self.Dn else None, keep,
                                        m.get("dnn"), [(self.lin_logit, 1.0), (self.cin_logit, 1.0)])
        return [(self.lin_logit, 1.0), (self.cin_logit, 1.0), (self.dnn_logit, 1.0)]

    def _branches_bwd(self, idx, dense, g, masks):
        p, gr = self.params, self.grads
        # DNN first: it STORES dLoss/dE into d_rows; every CIN layer then accumulates
        self.mlp.backward(g, self.d_rows.view(-1, self.FD), lin_grads=self._lin_grads())
        self._lin_done = self.mlp.lin_done
        ops.linear_dense_bwd(g, self.pooled, gr["cin_w"].view(-1), gr["cin_w0"], self.ws)
        cw = p["cin_w"].view(-1)
        L = len(self.units)
        drop = getattr(self, "_cin_drop", None)
        keep, cmasks, on = drop if drop else (None, None, [False] * (L + 1))
        X0 = self._cin_x0
        dX0 = self.d_rows
        if on[0]:  # dropped input: collect dLoss/d(dropped E) apart, then scale and add
            self._dx0_cin = torch.zeros_like(self.E)
            dX0 = self._dx0_cin
        for i in range(L - 1, -1, -1):
            n = self.units[i]
            pf, c0 = self.pool_from[i], self.pool_col0[i]
            d_hidden = self.dxk[i + 1] if i + 1 < L else None
            cwd, pfa = cw[c0: c0 + n - pf], pf
            if on[i + 1]:
                # dropped layer: the whole upstream gradient of its maps, built explicitly
                # ([next layer's dXk | g x cin_w]) and scaled by mask / keep, goes in as "hidden"
                up = torch.empty_like(self.maps[i])
                if pf:
                    up[:, :pf] = d_hidden
                up[:, pf:] = g.view(-1, 1, 1) * cw[c0: c0 + n - pf].view(1, -1, 1)  # broadcast over D
                up.mul_(cmasks[i + 1] / keep[i + 1])
                d_hidden, cwd, pfa = up, None, n
            ops.cin_layer_bwd(
                X0, X0 if i == 0 else self.maps[i - 1], self.Hs[i], p[f"cin_filter_{i}"][0],
                self.cin_act, self.maps[i], g, dX0, gr[f"cin_filter_{i}"][0],
                gr[f"cin_bias_{i}"], self.cin_bws, xk_is_x0=(i == 0),
                d_hidden=d_hidden, cin_w_direct=cwd, pool_from=pfa, accumulate_dx0=True,
                dXk=self.dxk[i] if i > 0 else None, split=self.hp.get("cin_gemm", "bf16x6") == "bf16x6",
                first6=self.hp.get("cin_first_layer", "bf16x6") == "bf16x6")
        if on[0]:
            self.d_rows.addcmul_(self._dx0_cin, cmasks[0] / keep[0])
        reg = self.hp.get("deep_l2_reg", 0.0)
        if reg:
            self.mlp.add_l2_grads(reg)
        reg = self.hp.get("cin_l2_reg", 0.0)
        if reg:
            for i in range(L):
                gr[f"cin_filter_{i}"].add_(p[f"cin_filter_{i}"], alpha=reg)
            gr["cin_w"].add_(p["cin_w"], alpha=reg)

    def _add_l2_model(self, total):
        reg = self.hp.get("deep_l2_reg", 0.0)
        if reg:
            total = total + self.mlp.l2(reg)
        reg = self.hp.get("cin_l2_reg", 0.0)
        if reg:
            ws = [self.params[f"cin_filter_{i}"] for i in range(len(self.units))] + [self.params["cin_w"]]
            total = total + sum(reg * 0.5 * w.square().sum() for w in ws)
        return total

    def roofline_probes(self, idx, dense, y):
        # the heaviest CIN forward layer on the f32 MFMA roofline
        B, m, D = idx.shape[0], self.F, self.D
        self._alloc(B)
        i = max(range(len(self.units)), key=lambda k: self.Hs[k] * self.units[k])
        H, n = self.Hs[i], self.units[i]
        xk = self.E if i == 0 else self.maps[i - 1]
        p = self.params
        self._embed(idx, dense, False, None)
        self._cin_fwd()

        def fn():
            return ops.cin_layer_fwd(self.E, xk, H, p[f"cin_filter_{i}"][0], p[f"cin_bias_{i}"],
                                     self.cin_act, self.maps[i], self.cin_fws, pooled=self.pooled,
                                     pool_col0=self.pool_col0[i], pool_from=self.pool_from[i], ws6=self.cin_fws6)

        flops = 2.0 * B * D * m * H * n
        if fn() is True:
            # the product path of this layer: fp32 operands as three bf16 pieces, six piece products per k-step
            # (csrc/cin6.hip) - priced on the bf16 pipe with the six products counted, the fp32 GEMM beside it
            return [dict(name=f"cin_fwd6_kernel (rm_cin_layer_fwd6, layer {i}: m={m} H={H} N={n}; Z = fl(x0 * xk) split "
                              "into 3 bf16 pieces, 6 piece products per k-step on the bf16 matrix pipe)",
                         symbol="cin_fwd6_kernel", fn=fn, work=6.0 * flops, bound="mfma", peak=2500.0,
                         extra=lambda ms: {"fp32_gemm_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
                                           "fp32_gemm_vs_f32_mfma_peak_157": round(flops / (ms * 1e-3) / 1e12 / 157.3, 4),
                                           "note": "peak = dense bf16 MFMA (2.5 PFLOP/s); achieved counts the six bf16 "
                                                   "products per fp32 product; the backward kernels (cin_dx / cin_dw: "
                                                   "f32 MFMA, 0.77-0.79 of 157.3 TFLOP/s) are now the longer launches"})]
        return [dict(name=f"cin_fwd_kernel (rm_cin_layer_fwd, layer {i}: m={m} H={H} N={n})",
                     symbol="cin_fwd_kernel", fn=fn, work=flops, bound="mfma")]


ENGINES = {"deepfm": DeepFMEngine, "dcn": DCNEngine, "xdeepfm": XDeepFMEngine}
