"""The reference's layer callables (recman/tf/core/layers.py) on the HIP kernels: the seam the
reference's models are composed at - `Layer(variables, ...)(tensor) -> tensor`, `layer.l2() -> scalar`,
variables upserted lazily by name into one shared dict (layers.py:95-110, 317-328, 532-574, 659-695,
789-794) - for users who build their own model the way xDeepFM._out does (xDeepFM.py:49-104,
DeepFM.py:107-163, DCN.py:99-149).  recman/th/layers.py, the file this fills, is 0 bytes in the reference.

    variables = {}
    emb = FeatEmbeddingLayer(variables, feat_dict, 16, l2, use_bias=False)
    E, _ = emb(inputs)                                   # inputs: th.inputs.DataInputs
    lin = SparseLinearLayer(variables, feats, l2)(SparseLinearCombiner(feats)(inputs))
    cin = CIN(variables, [128, 128], "leaky_relu", [1, 1, 1], l2)(E)
    dnn = DNN(variables, (32, 32), [1, 1, 1], "relu", l2)(DNNCombiner()([E] + inputs.dense_inputs(feat_dict)))
    pred = PredictionLayer(variables, "classification")(lin + cin + dnn)
    loss = create_loss(inputs.y, pred) + emb.l2() + ...
    loss.backward();  torch.optim.Adam(variables.values()).step()

Each layer is a torch.autograd.Function over recman_amd/ops.py (the C ABI, librecman_hip.so): forward and
backward are the same hand-written kernels the engines use (csrc/embed.hip, mlp.hip, gemm.hip, cin.hip,
cross.hip, loss.hip); torch is the plumbing in between - autograd's tape, the l2 terms, the bias-table
lookup and the loss on probabilities are plain torch ops.  Variables are float32 CUDA leaf tensors
under the reference's names, so `variables` is at once the parameter list of a torch optimizer and the
state dict of a checkpoint.  There is no CPU path.

This is the COMPOSABLE surface, not the fast one: every layer is its own launch group and the embedding
gradient comes back dense per feature (what TF's IndexedSlices + the dense l2 term add up to,
layers.py:188-193).  The model classes (th.DeepFM / DCN / xDeepFM) run the fused engines instead.
Supported features: SparseFeat and DenseFeat (multi-valued / value features: use the model classes).
CIN dropout is not offered here (keep-probabilities must be 1; the engines implement it)."""
import math

import numpy as np
import torch

from .. import engine as eng
from .. import ops
from .inputs import DenseFeat, MultiValCsvFeat, SparseFeat, SparseValueFeat

F32, I64 = torch.float32, torch.int64
KERAS_EPS = 1e-7


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("recman_amd.th.layers needs a GPU (MI355X); there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _leaf(t):
    return t.to(_device(), F32).contiguous().requires_grad_(True)


def glorot_normal(shape, seed=2019):
    """utils.py:156-183: truncated normal(0, sqrt(2 / (fan_in + fan_out))), +-2 sigma; fans as calc_fan
    computes them (the last two dimensions, times the product of the others)."""
    shape = tuple(int(s) for s in shape)
    k = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[-2] * k, shape[-1] * k
    std = math.sqrt(2.0 / (fan_in + fan_out))
    g = torch.Generator(device=_device()).manual_seed(int(seed))
    t = torch.empty(shape, device=_device(), dtype=F32)
    torch.nn.init.trunc_normal_(t, 0.0, std, -2 * std, 2 * std, generator=g)
    return t


def glorot_uniform(shape, seed=None):
    """utils.py:186-189 (unseeded in the reference, layers.py:690)."""
    shape = tuple(int(s) for s in shape)
    b = math.sqrt(6.0 / (shape[-2] + shape[-1]))
    g = None if seed is None else torch.Generator(device=_device()).manual_seed(int(seed))
    return torch.empty(shape, device=_device(), dtype=F32).uniform_(-b, b, generator=g)


def _dev_inputs(inputs, feat_dict=None):
    """(idx [B,F] int64, dense [B,Dn] f32) of a DataInputs on the GPU, cached on the object."""
    cached = getattr(inputs, "_dev_cache", None)
    if cached is None:
        idx = torch.from_numpy(np.ascontiguousarray(inputs.idx)).to(_device())
        dense = torch.from_numpy(np.ascontiguousarray(inputs.dense)).to(_device())
        cached = inputs._dev_cache = (idx, dense)
    return cached


def _keep_mask(shape, keep):
    """tf.nn.dropout(x, rate = 1 - keep): a 0/1 mask; kept entries are scaled by 1 / keep by the caller."""
    return (torch.rand(*shape, device=_device()) < keep).to(F32)


# ------------------------------------------------------------------------------------------------
# embedding gather
# ------------------------------------------------------------------------------------------------
class _GatherFn(torch.autograd.Function):
    """E[b,f,:] = table[field_off[f] + idx[b,f]] (rm_embed_fwd; backward rm_scatter_add_rows)."""

    @staticmethod
    def forward(ctx, table, idx, field_off):
        B, F = idx.shape
        D = table.shape[1]
        E = torch.empty(B, F, D, device=table.device, dtype=F32)
        ops.embed_fwd(idx, table.detach(), field_off, E=E, D=D, table_ld=D)
        ctx.save_for_backward(idx, field_off)
        ctx.R = table.shape[0]
        return E

    @staticmethod
    def backward(ctx, dE):
        idx, field_off = ctx.saved_tensors
        d_table = torch.zeros(ctx.R, dE.shape[2], device=dE.device, dtype=F32)
        ops.scatter_add_rows(d_table, idx, field_off, rows=dE.contiguous())
        return d_table, None, None


def _csr_dev(inputs, name):
    """(offsets, ids, vals or None) of a multi-valued / value feature on the GPU (DataInputs.mv holds the CSR)."""
    cache = inputs.__dict__.setdefault("_mv_dev_cache", {})
    if name not in cache:
        c = inputs.mv[name]
        dev = _device()
        cache[name] = (torch.from_numpy(np.ascontiguousarray(c.offsets)).to(dev),
                       torch.from_numpy(np.ascontiguousarray(c.ids)).to(dev),
                       None if c.vals is None else torch.from_numpy(np.ascontiguousarray(c.vals)).to(dev, F32))
    return cache[name]


class _PoolFn(torch.autograd.Function):
    """One multi-valued (MultiValCsvFeat: sqrtn combiner, layers.py:144-169) or value (SparseValueFeat: value *
    row, bias unscaled, layers.py:129-142) feature: (E_f [B,D], bias_f [B]) through rm_pool_rows over a temporary
    fused copy [V, D + 4] of the feature's table and bias table; backward rm_pool_rows_bwd."""

    @staticmethod
    def forward(ctx, table, bias, offsets, ids, vals):
        V, D = table.shape
        B = offsets.shape[0] - 1
        fused = torch.zeros(V, D + 4, device=table.device, dtype=F32)
        fused[:, :D] = table.detach()
        if bias is not None:
            fused[:, D] = bias.detach().reshape(-1)
        out = torch.empty(B, D + 4, device=table.device, dtype=F32)
        ops.pool_rows(fused, 0, D, offsets, ids, out, vals=vals)
        ctx.save_for_backward(offsets, ids, vals)
        ctx.shape, ctx.has_bias = (V, D), bias is not None
        return out[:, :D].contiguous(), out[:, D].contiguous()

    @staticmethod
    def backward(ctx, dE, dbias):
        offsets, ids, vals = ctx.saved_tensors
        V, D = ctx.shape
        d_table = torch.zeros(V, D, device=dE.device, dtype=F32)
        d_bias = torch.zeros(V, device=dE.device, dtype=F32) if ctx.has_bias else None
        ops.pool_rows_bwd(dE.contiguous(), dbias.contiguous() if ctx.has_bias else None, None, D, offsets, ids, 0,
                          d_table, d_bias, None, vals=vals)
        return d_table, (d_bias.view(V, 1) if ctx.has_bias else None), None, None, None


class FeatEmbedding:
    """layers.py:68-193: one feature's table (+ bias table).  SparseFeat (:117-128) features are gathered
    together by FeatEmbeddingLayer; MultiValCsvFeat (:144-169) and SparseValueFeat (:129-142) through _PoolFn."""

    display_name = "FeatEmbedding"

    def __init__(self, variables, feat, embedding_size, l2_reg=0.00001, use_bias=True, prefix="", seed=2019):
        assert not isinstance(feat, DenseFeat)  # layers.py:85
        if not isinstance(feat, (SparseFeat, MultiValCsvFeat, SparseValueFeat)):
            raise NotImplementedError(f"feature {feat.name}: th.layers embeds SparseFeat, MultiValCsvFeat and "
                                      "SparseValueFeat (the reference raises for the others too)")
        self.variables, self.feat, self.embedding_size = variables, feat, int(embedding_size)
        self.l2_reg, self.use_bias, self.prefix, self.seed = l2_reg, use_bias, prefix, seed

    def _upsert_variables(self):
        name = f"{self.prefix}{self.feat.name}_feat_embed"
        if name not in self.variables:
            self.variables[name] = _leaf(glorot_normal([self.feat.feat_size, self.embedding_size], self.seed))
        name = f"{self.prefix}{self.feat.name}_feat_bias"
        if name not in self.variables and self.use_bias:
            self.variables[name] = _leaf(torch.zeros(self.feat.feat_size, 1))

    def l2(self):
        return self.l2_reg * 0.5 * self.variables[f"{self.prefix}{self.feat.name}_feat_embed"].square().sum()


class FeatEmbeddingLayer:
    """layers.py:196-267: all embedding features -> (E [B,F,D], bias [B,F,1] or None), field order =
    feat_dict.embedding_feats.  The per-feature tables are the variables; they are concatenated for the
    one gather launch (a copy per call - the model classes keep one fused table instead)."""

    display_name = "FeatEmbeddingLayer"

    def __init__(self, variables, feat_dict, embedding_size, l2_reg=0.00001, use_bias=True, prefix="", seed=2019):
        if int(embedding_size) % 4:
            raise ValueError("embedding_size must be a multiple of 4 on the HIP path")
        self.variables, self.feat_dict, self.embedding_size = variables, feat_dict, int(embedding_size)
        self.l2_reg, self.use_bias, self.prefix, self.seed = l2_reg, use_bias, prefix, seed
        self.feat_embeds = dict((feat, FeatEmbedding(variables, feat, embedding_size, l2_reg, use_bias=use_bias,
                                                     prefix=prefix, seed=seed))
                                for feat in feat_dict.embedding_feats)

    def __call__(self, inputs):
        feats = list(self.feat_embeds)
        for fe in self.feat_embeds.values():
            fe._upsert_variables()
        idx, _ = _dev_inputs(inputs)
        if idx.shape[1] != len(feats):
            raise ValueError("inputs were not loaded with this feature dictionary")
        mv = getattr(inputs, "mv", None) or {}
        plain = [j for j, f in enumerate(feats) if f.name not in mv]
        cols_e, cols_b = [None] * len(feats), [None] * len(feats)
        if plain:  # the id features: ONE gather launch over their concatenated tables
            pf = [feats[j] for j in plain]
            sizes = [f.feat_size for f in pf]
            offs = torch.tensor(np.concatenate(([0], np.cumsum(sizes)[:-1])), dtype=I64, device=idx.device)
            table = torch.cat([self.variables[f"{self.prefix}{f.name}_feat_embed"] for f in pf], 0)
            pidx = idx if len(plain) == len(feats) else idx[:, plain].contiguous()
            Ep = _GatherFn.apply(table, pidx, offs)
            bp = None
            if self.use_bias:
                btab = torch.cat([self.variables[f"{self.prefix}{f.name}_feat_bias"] for f in pf], 0)
                bp = btab[(pidx + offs)]  # [B,Fp,1] (a torch lookup: B*Fp scalars)
            if len(plain) == len(feats):
                return Ep, bp
            for k, j in enumerate(plain):
                cols_e[j] = Ep[:, k]
                cols_b[j] = bp[:, k] if bp is not None else None
        for j, f in enumerate(feats):  # multi-valued / value features: pooled rows, one launch each
            if f.name in mv:
                offsets, ids, vals = _csr_dev(inputs, f.name)
                bt = self.variables[f"{self.prefix}{f.name}_feat_bias"] if self.use_bias else None
                e, b = _PoolFn.apply(self.variables[f"{self.prefix}{f.name}_feat_embed"], bt, offsets, ids, vals)
                cols_e[j], cols_b[j] = e, b.view(-1, 1)
        E = torch.stack(cols_e, 1)
        bias = torch.stack(cols_b, 1) if self.use_bias else None
        return E, bias

    def l2(self):
        return sum(fe.l2() for fe in self.feat_embeds.values())


# ------------------------------------------------------------------------------------------------
# linear term
# ------------------------------------------------------------------------------------------------
class _LinearInput:
    """What the combiners hand to the linear layers instead of a [B, sum V] one-hot matrix (26,000,039
    columns at the Criteo shape): the ids, their offsets into linear_w, the dense columns."""

    def __init__(self, idx, lin_off, dense, dense_pos, width):
        self.idx, self.lin_off, self.dense, self.dense_pos, self.width = idx, lin_off, dense, dense_pos, width
        self.shape = (idx.shape[0] if idx is not None else dense.shape[0], width)


class LinearCombiner:
    """layers.py:270-298 (and SparseLinearCombiner :357-386): the linear features in the order given;
    a sparse feature owns feat_size one-hot columns, a dense feature one column."""

    display_name = "LinearCombiner"

    def __init__(self, linear_feats, prefix=""):
        self.linear_feats, self.prefix = list(linear_feats), prefix

    def __call__(self, inputs):
        dev = _device()
        cols, offs, dcols, dpos, mvs = [], [], [], [], []
        at = 0
        for feat in self.linear_feats:
            if isinstance(feat, DenseFeat):
                dcols.append(np.asarray(inputs[feat.name], dtype=np.float32).reshape(-1, 1))
                dpos.append(at)
            elif isinstance(feat, (MultiValCsvFeat, SparseValueFeat)):
                # multi-hot count without slot 0 (utils.py:86-108) / value * one-hot (utils.py:70-71): weighted ids
                offsets, ids, vals = _csr_dev(inputs, feat.name)
                wts = vals if vals is not None else (ids >= 1).to(F32)
                seg = torch.repeat_interleave(torch.arange(offsets.shape[0] - 1, device=dev), offsets[1:] - offsets[:-1])
                mvs.append((at, ids, wts, seg))
            elif isinstance(feat, SparseFeat):
                cols.append(np.asarray(inputs[feat.name], dtype=np.int64).reshape(-1, 1))
                offs.append(at)
            else:
                raise NotImplementedError(f"feature {feat.name}: th.layers handles SparseFeat / SparseValueFeat / "
                                          "MultiValCsvFeat / DenseFeat")
            at += feat.feat_size
        idx = torch.from_numpy(np.concatenate(cols, 1)).to(dev) if cols else None
        dense = torch.from_numpy(np.concatenate(dcols, 1)).to(dev) if dcols else None
        self.output = _LinearInput(idx, torch.tensor(offs, dtype=I64, device=dev) if cols else None, dense,
                                   torch.tensor(dpos, dtype=I64, device=dev) if dcols else None, at)
        self.output.mvs = mvs
        return self.output


class SparseLinearCombiner(LinearCombiner):
    display_name = "SparseLinearCombiner"


class _LinearFn(torch.autograd.Function):
    """rm_linear_fwd; backward rm_scatter_add_rows (g_row form) + rm_linear_dense_bwd."""

    @staticmethod
    def forward(ctx, w, w0, idx, lin_off, dense, dense_pos):
        B = idx.shape[0] if idx is not None else dense.shape[0]
        wf = w.detach().reshape(-1)
        wd = wf[dense_pos].contiguous() if dense is not None else None
        out = torch.empty(B, device=w.device, dtype=F32)
        ops.linear_fwd(idx, lin_off, wf if idx is not None else None, dense, wd, w0.detach(), out)
        ctx.save_for_backward(idx, lin_off, dense, dense_pos)
        ctx.n = wf.numel()
        return out.view(B, 1)

    @staticmethod
    def backward(ctx, g):
        idx, lin_off, dense, dense_pos = ctx.saved_tensors
        g = g.reshape(-1).contiguous()
        d_w = torch.zeros(ctx.n, device=g.device, dtype=F32)
        if idx is not None:
            ops.scatter_add_rows(d_w, idx, lin_off, g_row=g, width=1, ld=1)
        ws = torch.empty(256 * 1024, device=g.device, dtype=F32)
        d_w0 = torch.empty(1, device=g.device, dtype=F32)
        Dn = 0 if dense is None else dense.shape[1]
        d_wd = torch.empty(Dn, device=g.device, dtype=F32) if Dn else None
        ops.linear_dense_bwd(g, dense, d_wd, d_w0, ws)
        if Dn:
            d_w[dense_pos] = d_wd
        return d_w.view(-1, 1), d_w0, None, None, None, None


class LinearLayer:
    """layers.py:301-354 (SparseLinearLayer :389-446): logit = one_hot(inputs) @ W + W0 in gather form;
    W, W0 zero-initialised (:317-328); training=False adds the features' manual weights (:338-345)."""

    display_name = "LinearRegression"

    def __init__(self, variables, linear_feats, l2_reg=0.00001, prefix="", training=True):
        self.variables, self.linear_feats = variables, list(linear_feats)
        self.l2_reg, self.prefix, self.training = l2_reg, prefix, training

    def _upsert_variables(self, input_shape):
        name = f"{self.prefix}linear_w0"
        if name not in self.variables:
            self.variables[name] = _leaf(torch.zeros(1))
        name = f"{self.prefix}linear_w"
        if name not in self.variables:
            self.variables[name] = _leaf(torch.zeros(input_shape[1], 1))

    def __call__(self, inputs):
        self._upsert_variables(inputs.shape)
        W = self.variables[f"{self.prefix}linear_w"]
        if not self.training:
            if any(getattr(f, "_weights", None) for f in self.linear_feats):
                mw = np.concatenate([np.asarray(f.weights, dtype=np.float64).reshape(-1) for f in self.linear_feats])
                W = W + torch.from_numpy(mw.astype(np.float32)).to(W.device).view(-1, 1)
        out = _LinearFn.apply(W, self.variables[f"{self.prefix}linear_w0"], inputs.idx, inputs.lin_off,
                              inputs.dense, inputs.dense_pos)
        for at, ids, wts, seg in getattr(inputs, "mvs", ()):  # multi-valued / value features: weighted ids
            part = torch.zeros(out.shape[0], device=out.device, dtype=F32)
            out = out + part.index_add(0, seg, W.view(-1)[at + ids] * wts).view(-1, 1)
        return out

    def l2(self):
        return self.l2_reg * 0.5 * self.variables[f"{self.prefix}linear_w"].square().sum()


class SparseLinearLayer(LinearLayer):
    display_name = "SparseLinearRegression"


# ------------------------------------------------------------------------------------------------
# FM
# ------------------------------------------------------------------------------------------------
class _FMFn(torch.autograd.Function):
    """FMLayer on a given E [B,F,D] (+ bias [B,F,1]): rm_embed_fwd with E itself as the table and the
    identity index (its gather + FM reduction is the FM layer); backward rm_embed_bwd."""

    @staticmethod
    def forward(ctx, E, bias, mask_b, mask_e):
        B, F, D = E.shape
        Ed = E.detach().contiguous()
        ident = torch.arange(B * F, device=E.device, dtype=I64).view(B, F)
        zoff = torch.zeros(F, device=E.device, dtype=I64)
        S = torch.empty(B, D, device=E.device, dtype=F32)
        out = torch.empty(B, device=E.device, dtype=F32)
        ops.embed_fwd(ident, Ed.view(B * F, D), zoff, D=D, table_ld=D,
                      bias_table=bias.detach().reshape(-1).contiguous() if bias is not None else None, bias_ld=1,
                      mask_b=mask_b, mask_e=mask_e, fm_sum=S, fm_logit=out)
        ctx.save_for_backward(Ed, S, mask_b, mask_e)
        ctx.has_bias = bias is not None
        return out.view(B, 1)

    @staticmethod
    def backward(ctx, g):
        Ed, S, mask_b, mask_e = ctx.saved_tensors
        B, F, D = Ed.shape
        g = g.reshape(-1).contiguous()
        dE = torch.empty_like(Ed)
        d_bias = torch.empty(B, F, device=g.device, dtype=F32)
        ops.embed_bwd(dE, E=Ed, fm_sum=S, g_fm=g, mask_b=mask_b, mask_e=mask_e, d_bias=d_bias)
        return dE, (d_bias.view(B, F, 1) if ctx.has_bias else None), None, None


class FMLayer:
    """layers.py:449-481: y = sum_f bias_f + 1/2 sum_k [(sum_f E_fk)^2 - sum_f E_fk^2]; `dropout` holds
    KEEP probabilities (bias, embeddings), rate = 1 - keep (:461,466)."""

    def __init__(self, dropout=(1, 1)):
        self.dropout = dropout

    def __call__(self, embeddings, embedding_bias):
        assert embeddings.dim() == 3  # layers.py:458
        B, F, D = embeddings.shape
        kb, ke = self.dropout
        mask_b = _keep_mask((B, F), kb) / kb if (kb < 1 and embedding_bias is not None) else None
        mask_e = _keep_mask((B, F, D), ke) / ke if ke < 1 else None
        return _FMFn.apply(embeddings, embedding_bias, mask_b, mask_e)

    def l2(self):
        return 0.0  # layers.py:480-481


# ------------------------------------------------------------------------------------------------
# DNN
# ------------------------------------------------------------------------------------------------
class _Concat:
    """x = [xe | xd] as the pair the kernels take (never concatenated in memory)."""

    def __init__(self, xe, xd):
        self.xe, self.xd = xe, xd
        self.shape = (xe.shape[0], xe.shape[1] + (0 if xd is None else xd.shape[1]))


class DNNCombiner:
    """layers.py:484-501: [flatten(E) | dense_1 .. dense_Dn]."""

    def __init__(self, prefix=""):
        self.prefix = prefix

    def __call__(self, inputs: list):
        E = inputs[0]
        xe = E.reshape(E.shape[0], -1)
        dense = [d if torch.is_tensor(d) else torch.from_numpy(np.asarray(d, dtype=np.float32)) for d in inputs[1:]]
        xd = torch.cat([d.reshape(xe.shape[0], -1).to(xe.device, F32) for d in dense], 1).contiguous() if dense else None
        return _Concat(xe, xd)


def _as_concat(x):
    if isinstance(x, _Concat):
        return x
    if x.shape[1] % 4:
        raise ValueError("a plain DNN / CrossNet input needs a multiple of 4 columns; pass DNNCombiner's output")
    return _Concat(x, None)


class _DNNFn(torch.autograd.Function):
    """engine.MLP (rm_mlp_fwd / rm_mlp_bwd fused for widths <= 32, rm_dense_fwd / rm_dense_wgrad otherwise)."""

    @staticmethod
    def forward(ctx, core, keep, masks, xe, xd, *params):
        out = core.mlp.forward(xe.detach().contiguous(), None if xd is None else xd.detach().contiguous(), keep, masks)
        ctx.core = core
        return out.clone().view(-1, 1)

    @staticmethod
    def backward(ctx, g):
        mlp = ctx.core.mlp
        dxe = torch.empty(g.shape[0], mlp.FD, device=g.device, dtype=F32)
        mlp.backward(g.reshape(-1).contiguous(), dxe)
        return (None, None, None, dxe, None) + tuple(mlp.g[n].clone() for n in ctx.core.names)


class DNN:
    """layers.py:504-628: x -> dropout -> [dense + bias + activation + dropout]* -> x @ dnn_w + dnn_w0.
    `dropout`: len(hidden_units) + 1 KEEP probabilities (:522); hidden_units=None -> the 2/3 rule of
    compute_hidden_units_s2 (utils.py:135-139)."""

    display_name = "DeepNeuralNetwork"

    def __init__(self, variables, hidden_units, dropout, activation, l2_reg=0.00001, prefix="", seed=2019):
        self.variables, self.hidden_units, self.dropout = variables, hidden_units, list(dropout)
        self.activation, self.l2_reg, self.prefix, self.seed = activation, l2_reg, prefix, seed
        self.mlp, self.names = None, None
        if hidden_units is not None:
            assert len(self.dropout) == len(hidden_units) + 1  # layers.py:522

    def _build(self, FD, Dn):
        hidden = self.hidden_units
        if hidden is None:  # utils.py:135-139
            h = (FD + Dn) * 2 // 3
            hidden = (h, h)
            assert len(self.dropout) == 3
        self.hidden = [int(h) for h in hidden]
        pre, dims = self.prefix, [FD + Dn] + self.hidden
        tmp_p, tmp_g = {}, {}
        self.mlp = eng.MLP(tmp_p, tmp_g, FD, Dn, self.hidden, self.activation, _device(), prefix=pre)
        self.names = list(tmp_p)
        for n in self.names:  # upsert under the reference's names and distributions (:532-574)
            if n not in self.variables:
                shape = tmp_p[n].shape
                init = glorot_normal(shape, self.seed) if (n.endswith("_weights") or n.endswith("dnn_w")) else torch.zeros(shape)
                self.variables[n] = _leaf(init)
        self.mlp.p = self.variables

    def __call__(self, x):
        x = _as_concat(x)
        FD, Dn = x.xe.shape[1], 0 if x.xd is None else x.xd.shape[1]
        if self.mlp is None or (self.mlp.FD, self.mlp.Dn) != (FD, Dn):
            self._build(FD, Dn)
        keep = self.dropout
        dims = [FD + Dn] + self.hidden
        masks = None
        if any(k < 1 for k in keep):
            masks = [_keep_mask((x.xe.shape[0], d), k) if k < 1 else None for d, k in zip(dims, keep)]
        return _DNNFn.apply(self, keep, masks, x.xe, x.xd, *[self.variables[n] for n in self.names])

    def l2(self):
        ws = [self.variables[f"{self.prefix}dnn_layer_{i}_weights"] for i in range(len(self.hidden))]
        ws.append(self.variables[f"{self.prefix}dnn_w"])
        return sum(self.l2_reg * 0.5 * w.square().sum() for w in ws)  # layers.py:611-628


# ------------------------------------------------------------------------------------------------
# CIN
# ------------------------------------------------------------------------------------------------
class _CINFn(torch.autograd.Function):
    """rm_cin_layer_fwd per layer + rm_rowdot; backward rm_cin_layer_bwd per layer (csrc/cin.hip, f32 MFMA).
    masks: None, or the L + 1 keep masks of cin_dropout (the input E, layers.py:707-708, then every layer's maps
    behind the activation, :740): a dropped layer's maps are scaled in place after its kernel and its pooled
    columns re-summed; in the backward its upstream gradient ([next layer's dXk | g x cin_w]) is built explicitly
    and scaled by mask / keep - as the xDeepFM engine does (engine.XDeepFMEngine._cin_fwd / _branches_bwd)."""

    @staticmethod
    def forward(ctx, core, masks, E, *params):
        c = core
        B, m, D = E.shape
        c._alloc(B, m, D)
        L = len(c.units)
        keep = c.dropout
        on = [bool(masks is not None and masks[i] is not None and keep[i] < 1) for i in range(L + 1)]
        X0 = E.detach().contiguous()
        if on[0]:
            X0 = X0 * (masks[0] / keep[0])
        xk = X0
        for i, n in enumerate(c.units):
            ops.cin_layer_fwd(X0, xk, c.Hs[i], c.variables[c.fname(i)].detach()[0], c.variables[c.bname(i)].detach(),
                              c.act, c.maps[i], c.fws, pooled=None if on[i + 1] else c.pooled,
                              pool_col0=c.pool_col0[i], pool_from=c.pool_from[i], ws6=c.fws6)
            if on[i + 1]:
                pf, c0 = c.pool_from[i], c.pool_col0[i]
                c.maps[i].mul_(masks[i + 1] / keep[i + 1])
                torch.sum(c.maps[i][:, pf:, :], dim=2, out=c.pooled[:, c0: c0 + n - pf])
            xk = c.maps[i]
        out = torch.empty(B, device=E.device, dtype=F32)
        ops.rowdot(c.pooled, c.variables[c.prefix + "cin_w"].detach().view(-1), c.variables[c.prefix + "cin_w0"].detach(), out)
        ctx.core, ctx.X0, ctx.masks, ctx.on = core, X0, masks, on
        return out.view(B, 1)

    @staticmethod
    def backward(ctx, g):
        c, X0, masks, on = ctx.core, ctx.X0, ctx.masks, ctx.on
        keep = c.dropout
        B, m, D = X0.shape
        g = g.reshape(-1).contiguous()
        L = len(c.units)
        dX0 = torch.zeros_like(X0)
        grads = {}
        d_cw, d_cw0 = torch.empty(c.P, device=g.device, dtype=F32), torch.empty(1, device=g.device, dtype=F32)
        ops.linear_dense_bwd(g, c.pooled, d_cw, d_cw0, c.ws)
        cw = c.variables[c.prefix + "cin_w"].detach().view(-1)
        for i in range(L - 1, -1, -1):
            n, pf, c0 = c.units[i], c.pool_from[i], c.pool_col0[i]
            dW = torch.empty(m * c.Hs[i], n, device=g.device, dtype=F32)
            db = torch.empty(n, device=g.device, dtype=F32)
            d_hidden = c.dxk[i + 1] if i + 1 < L else None
            cwd, pfa = cw[c0: c0 + n - pf], pf
            if on[i + 1]:
                up = torch.empty_like(c.maps[i])
                if pf:
                    up[:, :pf] = d_hidden
                up[:, pf:] = g.view(-1, 1, 1) * cw[c0: c0 + n - pf].view(1, -1, 1)
                up.mul_(masks[i + 1] / keep[i + 1])
                d_hidden, cwd, pfa = up, None, n
            ops.cin_layer_bwd(X0, X0 if i == 0 else c.maps[i - 1], c.Hs[i], c.variables[c.fname(i)].detach()[0], c.act,
                              c.maps[i], g, dX0, dW, db, c.bws, xk_is_x0=(i == 0),
                              d_hidden=d_hidden, cin_w_direct=cwd,
                              pool_from=pfa, accumulate_dx0=True, dXk=c.dxk[i] if i > 0 else None)
            grads[c.fname(i)], grads[c.bname(i)] = dW.view(1, m * c.Hs[i], n), db
        if on[0]:
            dX0 = dX0 * (masks[0] / keep[0])
        grads[c.prefix + "cin_w"], grads[c.prefix + "cin_w0"] = d_cw.view(-1, 1), d_cw0
        return (None, None, dX0) + tuple(grads[n] for n in c.names)


class CIN:
    """layers.py:631-777: per layer Z = X0 (x) Xk (never formed), 1x1 convolution with the layer's filter,
    activation BEFORE the split, first half -> next hidden / second half -> direct (last layer: all
    direct), sum over D, @ cin_w + cin_w0.  Non-final layer sizes must be even (the reference silently
    drops a map otherwise, :681,742)."""

    display_name = "CompressedInteractionNetwork"

    def __init__(self, variables, cross_layer_units, activation, dropout, l2_reg=0.00001, prefix="", seed=2019):
        self.variables, self.cross_layer_units = variables, [int(u) for u in cross_layer_units]
        self.activation, self.dropout, self.l2_reg, self.prefix, self.seed = activation, list(dropout), l2_reg, prefix, seed
        assert len(self.cross_layer_units) > 0  # layers.py:656
        assert len(self.cross_layer_units) + 1 == len(self.dropout)  # :657
        self.units, self.act = self.cross_layer_units, eng.act_name(activation)
        self._shape = None

    def fname(self, i):
        return f"{self.prefix}cin_filter_{i}"

    def bname(self, i):
        return f"{self.prefix}cin_bias_{i}"

    def _upsert_variables(self, field_size):
        m = field_size
        self.Hs, self.pool_from, self.pool_col0 = [m], [], []
        final = 0
        for i, size in enumerate(self.units):
            last = i == len(self.units) - 1
            if not last and size % 2:
                raise ValueError("CIN layer sizes before the last must be even (split in halves, layers.py:742-746)")
            H = self.Hs[-1]
            if self.fname(i) not in self.variables:  # :663-676
                self.variables[self.fname(i)] = _leaf(glorot_normal([1, m * H, size], self.seed))
            if self.bname(i) not in self.variables:
                self.variables[self.bname(i)] = _leaf(torch.zeros(size))
            self.pool_from.append(0 if last else size // 2)
            self.pool_col0.append(final)
            final += size if last else size // 2
            self.Hs.append(size // 2)
        self.P = final
        if self.prefix + "cin_w" not in self.variables:  # :687-695 (glorot_uniform, unseeded)
            self.variables[self.prefix + "cin_w"] = _leaf(glorot_uniform([final, 1]))
        if self.prefix + "cin_w0" not in self.variables:
            self.variables[self.prefix + "cin_w0"] = _leaf(torch.zeros(1))
        self.names = ([n for i in range(len(self.units)) for n in (self.fname(i), self.bname(i))]
                      + [self.prefix + "cin_w", self.prefix + "cin_w0"])

    def _alloc(self, B, m, D):
        if self._shape == (B, m, D):
            return
        self._shape = (B, m, D)
        dev = _device()
        self.maps = [torch.empty(B, n, D, dtype=F32, device=dev) for n in self.units]
        self.dxk = [None] + [torch.empty(B, self.Hs[i], D, dtype=F32, device=dev) for i in range(1, len(self.units))]
        self.pooled = torch.empty(B, self.P, dtype=F32, device=dev)
        fw = max(ops.cin_filter_workspace(m, self.Hs[i], n) for i, n in enumerate(self.units))
        bw = max(ops.cin_bwd_workspace(B, m, self.Hs[i], n, D) for i, n in enumerate(self.units))
        self.fws = torch.empty(fw, dtype=F32, device=dev)
        self.bws = torch.empty(bw, dtype=F32, device=dev)
        f6 = max(ops.cin_filter_workspace6(m, self.Hs[i], n, D) for i, n in enumerate(self.units))
        self.fws6 = torch.empty(f6, dtype=F32, device=dev) if f6 > 0 else None  # (rm_cin_layer_fwd6, csrc/cin6.hip)
        self.ws = torch.empty(256 * 1024, dtype=F32, device=dev)

    def __call__(self, inputs):
        assert inputs.dim() == 3  # layers.py:698
        self._upsert_variables(inputs.shape[1])
        masks = None
        if any(k < 1 for k in self.dropout):  # tf.nn.dropout on the input (:707-708) and on every layer's maps (:740)
            B, _, D = inputs.shape
            shapes = [tuple(inputs.shape)] + [(B, n, D) for n in self.units]
            masks = [_keep_mask(sh, k) if k < 1 else None for sh, k in zip(shapes, self.dropout)]
        return _CINFn.apply(self, masks, inputs, *[self.variables[n] for n in self.names])

    def l2(self):
        ws = [self.variables[self.fname(i)] for i in range(len(self.units))] + [self.variables[self.prefix + "cin_w"]]
        return sum(self.l2_reg * 0.5 * w.square().sum() for w in ws)  # layers.py:762-777


# ------------------------------------------------------------------------------------------------
# CrossNet
# ------------------------------------------------------------------------------------------------
class _CrossFn(torch.autograd.Function):
    """rm_cross_fwd / rm_cross_bwd (all layers fused) + rm_dense_wgrad / rm_cross_param_grads for the
    parameter gradients, as the DCN engine runs them."""

    @staticmethod
    def forward(ctx, core, xe, xd, w, b, w_out):
        B = xe.shape[0]
        L = w.shape[0]
        xe_, xd_ = xe.detach().contiguous(), None if xd is None else xd.detach().contiguous()
        logit = torch.empty(B, device=xe.device, dtype=F32)
        p = torch.empty(B, ops.cross_p_ld(L), device=xe.device, dtype=F32)
        ops.cross_fwd(xe_, xd_, w.detach(), b.detach(), w_out.detach().view(-1), logit, p)
        ctx.save_for_backward(xe_, xd_, w.detach(), b.detach(), w_out.detach(), p)
        return logit.view(B, 1)

    @staticmethod
    def backward(ctx, g):
        xe, xd, w, b, w_out, p = ctx.saved_tensors
        B, FD = xe.shape
        L, d = w.shape
        g = g.reshape(-1).contiguous()
        dev = g.device
        d_xe = torch.empty(B, FD, device=dev, dtype=F32)
        coef = torch.empty(B, 2 * L + 2, device=dev, dtype=F32)
        ops.cross_bwd(w, b, w_out.view(-1), g, p, d_xe, coef)
        P = torch.empty(d, L + 1, device=dev, dtype=F32)
        wws = torch.empty(max(1, ops.dense_wgrad_workspace(d, L + 1, B)), device=dev, dtype=F32)
        ops.dense_wgrad(xe, xd, coef[:, : L + 1], P, wws)
        csum = torch.empty(2 * L + 2, device=dev, dtype=F32)
        ops.linear_dense_bwd(torch.ones(B, device=dev, dtype=F32), coef, csum, None,
                             torch.empty(256 * 1024, device=dev, dtype=F32))
        dw, db, dwo = torch.empty_like(w), torch.empty_like(b), torch.empty(d, device=dev, dtype=F32)
        ops.cross_param_grads(P, csum[L + 1:].contiguous(), w, b, w_out.view(-1), dw, db, dwo)
        return None, d_xe, None, dw, db, dwo.view(-1, 1)


class CrossNet:
    """The class DCN.py:7 imports commented out and DCN.py:134-137 uses: CrossNet(layer_num, l2_reg)(x) ->
    logit [B,1], `.weights`, `.l2()`.  Arithmetic: arXiv 1708.05123 eq. (3) (the paper README.md:6 cites),
    x_{l+1} = x0 (x_l . w_l) + b_l + x_l, logit = x_L . w_out.  Variables (names chosen here - the
    reference has none): cross_w [L,d], cross_b [L,d], cross_w_out [d,1]."""

    display_name = "CrossNet"

    def __init__(self, variables, layer_num, l2_reg=0.0, prefix="", seed=2019):
        self.variables, self.layer_num, self.l2_reg, self.prefix, self.seed = variables, int(layer_num), l2_reg, prefix, seed

    def _upsert_variables(self, d):
        L, pre = self.layer_num, self.prefix
        if pre + "cross_w" not in self.variables:
            self.variables[pre + "cross_w"] = _leaf(glorot_normal([L, d, 1], self.seed).view(L, d))
        if pre + "cross_b" not in self.variables:
            self.variables[pre + "cross_b"] = _leaf(torch.zeros(L, d))
        if pre + "cross_w_out" not in self.variables:
            self.variables[pre + "cross_w_out"] = _leaf(glorot_normal([d, 1], self.seed))

    @property
    def weights(self):
        return [self.variables[self.prefix + n] for n in ("cross_w", "cross_b", "cross_w_out")]

    def __call__(self, x):
        x = _as_concat(x)
        self._upsert_variables(x.shape[1])
        w, b, wo = self.weights
        return _CrossFn.apply(self, x.xe, x.xd, w, b, wo)

    def l2(self):
        return self.l2_reg * 0.5 * (self.variables[self.prefix + "cross_w"].square().sum()
                                    + self.variables[self.prefix + "cross_w_out"].square().sum())


# ------------------------------------------------------------------------------------------------
# prediction + loss
# ------------------------------------------------------------------------------------------------
class _SigmoidFn(torch.autograd.Function):
    """PredictionLayer's sigmoid through rm_logit_loss (csrc/loss.hip)."""

    @staticmethod
    def forward(ctx, z):
        zf = z.detach().reshape(-1).contiguous()
        pred, logit = torch.empty_like(zf), torch.empty_like(zf)
        ops.logit_loss([(zf, 1.0)], task="classification", logit=logit, pred=pred)
        ctx.save_for_backward(pred)
        return pred

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        return (g * p * (1 - p)).view(-1, 1)


class PredictionLayer:
    """layers.py:780-808: optional global bias, sigmoid for classification, reshape to [B]."""

    display_name = "Prediction"

    def __init__(self, variables, task="classification", use_bias=False, prefix=""):
        self.task, self.use_bias, self.prefix, self.variables = task, use_bias, prefix, variables

    def __call__(self, inputs):
        output = inputs.reshape(-1, 1)
        if self.use_bias:
            name = f"{self.prefix}global_bias"
            if name not in self.variables:
                self.variables[name] = _leaf(torch.zeros(1))
            output = output + self.variables[name]
        if self.task == "classification":
            return _SigmoidFn.apply(output)
        return output.reshape(-1)


def create_loss(y, pred, task="classification"):
    """create_loss (utils.py:192-198): Keras binary_crossentropy on PROBABILITIES (clip to
    [1e-7, 1 - 1e-7], epsilon inside the logs), mean over the batch; or MSE."""
    y = torch.as_tensor(np.asarray(y) if not torch.is_tensor(y) else y).to(pred.device, pred.dtype).reshape(-1)
    if task == "classification":
        pc = pred.clamp(KERAS_EPS, 1 - KERAS_EPS)
        return (-(y * torch.log(pc + KERAS_EPS) + (1 - y) * torch.log(1 - pc + KERAS_EPS))).mean()
    if task == "regression":
        return (pred - y).square().mean()
    raise ValueError(f"unknown task {task!r}")  # utils.py:198


__all__ = ["FeatEmbedding", "FeatEmbeddingLayer", "LinearCombiner", "LinearLayer", "SparseLinearCombiner",
           "SparseLinearLayer", "FMLayer", "DNNCombiner", "DNN", "CIN", "CrossNet", "PredictionLayer",
           "create_loss", "glorot_normal", "glorot_uniform"]
