"""CPU PyTorch restatement (float32 or float64, autograd) of the reference layers
and of the three model compositions.

TEST INFRASTRUCTURE - see oracle/__init__.py.  This is "recman's own CPU PyTorch
path" of the north star: the reference's recman/th is an empty stub, so the
arithmetic follows recman/tf/core/layers.py (cited per function).  It is the
parity checker for the HIP path and the timed CPU baseline of bench.py.

Written independently of oracle/np_layers.py (gather form instead of one-hot,
einsum instead of split/matmul/reshape) so the two can check each other.
"""
import math

import torch
import torch.nn.functional as F_

KERAS_EPS = 1e-7


def act_fn(name):
    if callable(name):
        return name
    if name in (None, "linear", "identity"):
        return lambda x: x
    if name == "relu":
        return torch.relu
    if name == "leaky_relu":  # tf.nn.leaky_relu default alpha=0.2 (hparams/xDeepFM.py:29,33)
        return lambda x: F_.leaky_relu(x, 0.2)
    if name == "sigmoid":
        return torch.sigmoid
    if name == "tanh":
        return torch.tanh
    raise ValueError(name)


def dropout(x, keep, mask=None):
    """tf.nn.dropout(x, rate=1-keep) with an injected 0/1 keep mask."""
    if keep >= 1 or mask is None:
        return x
    return x * mask / keep


class Spec:
    """Which columns of idx / dense are which features (host-side bookkeeping).

    sparse_names / feat_sizes: embedding features in FeatureDictionary order
    (recman/tf/inputs.py:13-15); feat_size already includes the null slot 0
    (inputs.py:166).  dense_names: DenseFeat columns in dictionary order.
    """

    def __init__(self, sparse_names, feat_sizes, dense_names=(), multi_names=(), value_names=(),
                 linear_names=None):
        # linear_names: the hyper-parameter `linear_features` (utils.py:27-30) as a list - the features
        # of the linear term IN THE GIVEN ORDER (any mix of embedding and dense features); None = all of
        # them in the default order of utils.py:31-36
        self.linear_names = list(linear_names) if linear_names else None
        self.sparse_names = list(sparse_names)
        self.feat_sizes = [int(v) for v in feat_sizes]
        self.dense_names = list(dense_names)
        # embedding features that are MultiValCsvFeat (inputs.py:380-425): their idx column is
        # ignored, their tag ids come as CSR (offsets [B+1], ids [nnz]) in the `mv` dict
        self.multi_names = list(multi_names)
        # embedding features that are SparseValueFeat (inputs.py:213-278): idx column ignored,
        # `mv[name]` = (ids [B], vals [B])
        self.value_names = list(value_names)

    @property
    def F(self):
        return len(self.sparse_names)

    @property
    def Dn(self):
        return len(self.dense_names)

    @property
    def lin_offsets(self):
        # one-hot block offsets of LinearCombiner (layers.py:284-293) with the feature order
        # of utils.py:31-36: sparse feats first, then value feats, then multi-valued csv feats,
        # dense last.  Returned per embedding feature IN sparse_names ORDER, plus the dense
        # block start.
        size = dict(zip(self.sparse_names, self.feat_sizes))
        special = set(self.multi_names) | set(self.value_names)
        order = ([n for n in self.sparse_names if n not in special]
                 + [n for n in self.sparse_names if n in self.value_names]
                 + [n for n in self.sparse_names if n in self.multi_names])
        off, at = 0, {}
        for n in order:
            at[n] = off
            off += size[n]
        return [at[n] for n in self.sparse_names], off

    @property
    def lin_layout(self):
        """(offset of every embedding feature's one-hot block or None, offset of every dense
        feature's column or None, total width of linear_w) under `linear_names`."""
        if self.linear_names is None:
            offs, dense_off = self.lin_offsets
            return offs, [dense_off + j for j in range(self.Dn)], dense_off + self.Dn
        size = dict(zip(self.sparse_names, self.feat_sizes))
        at, off = {}, 0
        for n in self.linear_names:
            if n in at or (n not in size and n not in self.dense_names):
                raise ValueError(f"linear_features: unknown or repeated feature {n!r}")
            at[n] = off
            off += size.get(n, 1)
        return ([at.get(n) for n in self.sparse_names], [at.get(n) for n in self.dense_names], off)


# ---------------------------------------------------------------------------
# parameter creation: names and shapes of the reference variables
# ---------------------------------------------------------------------------
def _trunc_normal(shape, std, gen, dtype):
    t = torch.empty(shape, dtype=dtype)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
    return t


def glorot_std(shape):
    # calc_fan + glorot_normal (utils.py:156-165, 180-183)
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        k = 1
        for s in shape[:-2]:
            k *= s
        fan_in, fan_out = shape[-2] * k, shape[-1] * k
    return math.sqrt(2.0 / (fan_in + fan_out))


def make_params(spec, model, D, hidden=(32, 32), cin_units=(), cross_layers=0, use_bias=True,
                seed=2019, dtype=torch.float32, scale=None):
    """Variables with the reference's names (layers.py:96,106,318,324,533,541,548,
    558,564,572,663,673,687,693).  Values: truncated normal with the glorot std of
    the reference initialiser, or N(0, scale) when `scale` is given (the bench uses
    0.01); zero-initialised variables of the reference (bias tables, linear_w,
    biases) are made non-zero too so that parity tests exercise them.  TF's RNG
    cannot be reproduced, so parity tests always inject these tensors into both sides.
    """
    g = torch.Generator().manual_seed(seed)

    def rnd(shape):
        std = scale if scale is not None else glorot_std(shape)
        return _trunc_normal(shape, std, g, dtype)

    p = {}
    for name, V in zip(spec.sparse_names, spec.feat_sizes):
        p[f"{name}_feat_embed"] = rnd((V, D))
        if use_bias:
            p[f"{name}_feat_bias"] = rnd((V, 1))
    p["linear_w"] = rnd((spec.lin_layout[2], 1))
    p["linear_w0"] = rnd((1,))
    d_in = spec.F * D + spec.Dn
    dims = [d_in] + list(hidden)
    for i in range(len(hidden)):
        p[f"dnn_layer_{i}_weights"] = rnd((dims[i], dims[i + 1]))
        p[f"dnn_layer_{i}_bias"] = rnd((dims[i + 1],))
    p["dnn_w"] = rnd((dims[-1], 1))
    p["dnn_w0"] = rnd((1,))
    if cin_units:
        field_nums, final = [spec.F], 0
        for i, size in enumerate(cin_units):
            p[f"cin_filter_{i}"] = rnd((1, field_nums[-1] * field_nums[0], size))
            p[f"cin_bias_{i}"] = rnd((size,))
            field_nums.append(size // 2)
            final += field_nums[-1] if i != len(cin_units) - 1 else size
        p["cin_w"] = rnd((final, 1))
        p["cin_w0"] = rnd((1,))
    if cross_layers:
        p["cross_w"] = rnd((cross_layers, d_in))
        p["cross_b"] = rnd((cross_layers, d_in))
        p["cross_w_out"] = rnd((d_in, 1))
    return p


# ---------------------------------------------------------------------------
# layers
# ---------------------------------------------------------------------------
SPARSE_GRAD = False  # bench.py's CPU baseline: IndexedSlices-like sparse table gradients


def _lookup(table, ids):
    if SPARSE_GRAD:
        return F_.embedding(ids, table, sparse=True)
    return table[ids]


def pooled_lookup(table, offsets, ids):
    """tf.nn.embedding_lookup_sparse(table, sp_ids, None, combiner="sqrtn") (layers.py:150-156):
    per example sum of the looked-up rows / sqrt(number of ids); no ids -> zeros."""
    B = offsets.shape[0] - 1
    n = offsets[1:] - offsets[:-1]
    seg = torch.repeat_interleave(torch.arange(B), n)
    out = torch.zeros(B, table.shape[1], dtype=table.dtype).index_add(0, seg, table[ids])
    return out / n.clamp(min=1).to(table.dtype).sqrt().unsqueeze(1)


def feat_embedding_layer(p, spec, idx, use_bias=True, mv=None):
    """FeatEmbeddingLayer.__call__ (layers.py:238-261) over FeatEmbedding.__call__:
    SparseFeat branch (layers.py:117-128) and the sqrtn-pooled MultiValCsvFeat branch
    (layers.py:144-169): E [B,F,D], bias [B,F,1] or None.  SparseValueFeat branch
    (layers.py:129-142): embedding row * value, bias row unscaled - the INTENDED per-example
    scaling (the reference's `tf.multiply([B,1,D], [B])` cannot run for B != D)."""
    def one(suffix, f, n):
        t = p[f"{n}_feat_{suffix}"]
        if n in spec.multi_names:
            return pooled_lookup(t, *mv[n])
        if n in spec.value_names:
            ids, vals = mv[n]
            rows = _lookup(t, ids)
            return rows * vals.to(t.dtype).unsqueeze(1) if suffix == "embed" else rows
        return _lookup(t, idx[:, f])

    E = torch.stack([one("embed", f, n) for f, n in enumerate(spec.sparse_names)], dim=1)
    bias = None
    if use_bias:
        bias = torch.stack([one("bias", f, n) for f, n in enumerate(spec.sparse_names)], dim=1)
    return E, bias


def embedding_l2(p, spec, l2_reg):
    return sum(l2_reg * 0.5 * p[f"{n}_feat_embed"].square().sum() for n in spec.sparse_names)


def linear_layer(p, spec, idx, dense, manual_weights=None, mv=None):
    """LinearCombiner + LinearLayer (layers.py:281-347) / the Sparse* twins
    (layers.py:368-439) in gather form: one_hot(idx) @ W == W[offset + idx]; a
    MultiValCsvFeat contributes its multi-hot counts with slot 0 zeroed (utils.py:86-108)."""
    W = p["linear_w"]
    if manual_weights is not None:  # training=False (layers.py:338-345, 426-437)
        W = W + manual_weights.reshape(-1, 1).to(W.dtype)
    offs, dense_offs, _ = spec.lin_layout
    out = p["linear_w0"].reshape(1, 1).expand(idx.shape[0], 1)
    for f, off in enumerate(offs):
        if off is None:  # not among the linear_features
            continue
        n = spec.sparse_names[f]
        if n in spec.multi_names:
            offsets, ids = mv[n]
            B = offsets.shape[0] - 1
            seg = torch.repeat_interleave(torch.arange(B), offsets[1:] - offsets[:-1])
            contrib = W[off + ids] * (ids >= 1).to(W.dtype).unsqueeze(1)
            out = out + torch.zeros(B, 1, dtype=W.dtype).index_add(0, seg, contrib)
            continue
        if n in spec.value_names:  # one_hot(id) * value (utils.py:70-71)
            ids, vals = mv[n]
            out = out + _lookup(W, off + ids) * vals.to(W.dtype).unsqueeze(1)
            continue
        out = out + _lookup(W, off + idx[:, f])
    for j, off in enumerate(dense_offs):
        if off is not None:
            out = out + dense[:, j : j + 1] * W[off]
    return out


def linear_l2(p, l2_reg):
    return l2_reg * 0.5 * p["linear_w"].square().sum()


def fm_layer(E, bias, keep=(1, 1), masks=(None, None)):
    """FMLayer.__call__ (layers.py:457-478)."""
    bias = dropout(bias, keep[0], masks[0])
    y1 = bias.sum(dim=1)  # [B,1]
    E = dropout(E, keep[1], masks[1])
    s = E.sum(dim=1)  # [B,D]
    y2 = 0.5 * (s.square() - E.square().sum(dim=1)).sum(dim=1, keepdim=True)
    return y1 + y2


def dnn_input(E, dense):
    """DNNCombiner (layers.py:494-501): [flatten(E) | dense columns]."""
    x = E.reshape(E.shape[0], -1)
    if dense is not None and dense.shape[1]:
        x = torch.cat([x, dense], dim=1)
    return x


def dnn(p, x, n_layers, activation="relu", keep=None, masks=None):
    """DNN.__call__ (layers.py:576-609)."""
    keep = keep or [1] * (n_layers + 1)
    masks = masks or [None] * (n_layers + 1)
    act = act_fn(activation)
    y = dropout(x, keep[0], masks[0])
    for i in range(n_layers):
        y = act(y @ p[f"dnn_layer_{i}_weights"] + p[f"dnn_layer_{i}_bias"])
        y = dropout(y, keep[i + 1], masks[i + 1])
    return y @ p["dnn_w"] + p["dnn_w0"]


def dnn_l2(p, n_layers, l2_reg):
    ws = [p[f"dnn_layer_{i}_weights"] for i in range(n_layers)] + [p["dnn_w"]]
    return sum(l2_reg * 0.5 * w.square().sum() for w in ws)


def cin(p, E, n_layers, activation="leaky_relu", keep=None, masks=None, return_maps=False):
    """CIN.__call__ (layers.py:697-760).  Z[b,d,i*H+j] = X0[b,i,d] * Xk[b,j,d]
    (X0's field index major, layers.py:721-726), feature map = Z @ filter + bias,
    activation BEFORE the split, first half -> next hidden, second half -> direct
    (layers.py:738-749), pooled over D, @ cin_w + cin_w0."""
    B, m, D = E.shape
    keep = keep or [1] * (n_layers + 1)
    masks = masks or [None] * (n_layers + 1)
    act = act_fn(activation)
    x0 = dropout(E, keep[0], masks[0])
    xk = x0
    finals, maps = [], []
    for i in range(n_layers):
        W = p[f"cin_filter_{i}"][0]  # [m*H, N]
        N = W.shape[1]
        Z = torch.einsum("bid,bjd->bdij", x0, xk).reshape(B, D, -1)
        fm_ = act(Z @ W + p[f"cin_bias_{i}"])  # [B,D,N]
        fm_ = fm_.transpose(1, 2)  # [B,N,D]
        fm_ = dropout(fm_, keep[i + 1], masks[i + 1])
        maps.append(fm_)
        half = N // 2
        if i != n_layers - 1:
            xk, direct = fm_[:, :half], fm_[:, half : 2 * half]
        else:
            direct = fm_
        finals.append(direct)
    pooled = torch.cat(finals, dim=1).sum(dim=-1)
    logit = pooled @ p["cin_w"] + p["cin_w0"]
    if return_maps:
        return logit, pooled, maps
    return logit


def cin_l2(p, n_layers, l2_reg):
    ws = [p[f"cin_filter_{i}"] for i in range(n_layers)] + [p["cin_w"]]
    return sum(l2_reg * 0.5 * w.square().sum() for w in ws)


def cross_net(p, x0):
    """CrossNet is ABSENT from the reference (DCN.py:7,134-137).  cross_w [L,d]: DCN-v1 vector
    form, arXiv 1708.05123 eq. (3): x_{l+1} = x0 * (x_l . w_l) + b_l + x_l.  cross_w [L,d,d]: the
    matrix form (DCN-v2, arXiv 2008.13535 eq. (1)): x_{l+1} = x0 o (W_l x_l + b_l) + x_l.
    logit = x_L @ w_out."""
    x = x0
    W = p["cross_w"]
    for l in range(W.shape[0]):
        if W.dim() == 3:
            x = x0 * (x @ W[l].t() + p["cross_b"][l]) + x
        else:
            s = (x * W[l]).sum(dim=1, keepdim=True)
            x = x0 * s + p["cross_b"][l] + x
    return x @ p["cross_w_out"]


def cross_l2(p, l2_reg):
    return l2_reg * 0.5 * (p["cross_w"].square().sum() + p["cross_w_out"].square().sum())


def prediction(logit, task="classification"):
    """PredictionLayer (layers.py:796-808), use_bias=False as every model passes."""
    out = torch.sigmoid(logit) if task == "classification" else logit
    return out.reshape(-1)


def create_loss(y, pred, task="classification"):
    """create_loss (utils.py:192-198): Keras binary_crossentropy on PROBABILITIES
    (clip to [1e-7, 1-1e-7], log(p + 1e-7)), mean over the batch; or MSE."""
    y = y.to(pred.dtype)
    if task == "classification":
        pc = pred.clamp(KERAS_EPS, 1 - KERAS_EPS)
        bce = y * torch.log(pc + KERAS_EPS) + (1 - y) * torch.log(1 - pc + KERAS_EPS)
        return (-bce).mean()
    return (pred - y).square().mean()


# ---------------------------------------------------------------------------
# model compositions
# ---------------------------------------------------------------------------
def deepfm_logit(p, spec, idx, dense, hp, training=True, masks=None, manual_weights=None, mv=None):
    """DeepFM._init_graph (DeepFM.py:107-158): final = linear + fm + dnn."""
    masks = masks or {}
    E, bias = feat_embedding_layer(p, spec, idx, use_bias=True, mv=mv)
    logit = linear_layer(p, spec, idx, dense, manual_weights, mv)
    if hp.get("use_fm", True):
        keep = hp.get("fm_dropout", (1, 1)) if training else (1, 1)
        logit = logit + fm_layer(E, bias, keep, masks.get("fm", (None, None)))
    if hp.get("use_deep", True):
        n = len(hp["deep_hidden_units"])
        keep = hp.get("deep_dropout", [1] * (n + 1)) if training else [1] * (n + 1)
        logit = logit + dnn(p, dnn_input(E, dense), n, hp.get("deep_activation", "relu"),
                            keep, masks.get("dnn"))
    return logit


def deepfm_l2(p, spec, hp):
    out = embedding_l2(p, spec, hp.get("embedding_l2_reg", 0.0)) + linear_l2(
        p, hp.get("linear_l2_reg", 0.0))
    if hp.get("use_deep", True):
        out = out + dnn_l2(p, len(hp["deep_hidden_units"]), hp.get("deep_l2_reg", 0.0))
    return out  # DeepFM.py:164-180 (FMLayer.l2 is 0)


def dcn_logit(p, spec, idx, dense, hp, training=True, masks=None, manual_weights=None, mv=None):
    """DCN._init_graph (DCN.py:99-144): dnn_input feeds DNN and CrossNet;
    final = dnn + cross + dnn (dnn counted twice, DCN.py:140-142) only with
    strict_reference, else dnn + cross; + linear if use_linear."""
    masks = masks or {}
    E, _ = feat_embedding_layer(p, spec, idx, use_bias=False, mv=mv)
    x = dnn_input(E, dense)
    n = len(hp["deep_hidden_units"])
    keep = hp.get("deep_dropout", [1] * (n + 1)) if training else [1] * (n + 1)
    dnn_logit = dnn(p, x, n, hp.get("deep_activation", "relu"), keep, masks.get("dnn"))
    logit = dnn_logit + cross_net(p, x)
    if hp.get("strict_reference", False):
        logit = logit + dnn_logit
    if hp.get("use_linear", True):
        logit = logit + linear_layer(p, spec, idx, dense, manual_weights, mv)
    return logit


def dcn_l2(p, spec, hp):
    out = embedding_l2(p, spec, hp.get("embedding_l2_reg", 0.0))
    if hp.get("use_linear", True):
        out = out + linear_l2(p, hp.get("linear_l2_reg", 0.0))
    out = out + dnn_l2(p, len(hp["deep_hidden_units"]), hp.get("deep_l2_reg", 0.0))
    return out + cross_l2(p, hp.get("cross_layer_l2_reg", 0.0))  # DCN.py:151-166


def xdeepfm_logit(p, spec, idx, dense, hp, training=True, masks=None, manual_weights=None, mv=None):
    """xDeepFM._out (xDeepFM.py:47-104): embeddings without bias tables,
    final = linear + cin + dnn."""
    masks = masks or {}
    E, _ = feat_embedding_layer(p, spec, idx, use_bias=False, mv=mv)
    logit = linear_layer(p, spec, idx, dense, manual_weights, mv)
    nc = len(hp["cin_cross_layer_units"])
    keep = hp.get("cin_dropout", [1] * (nc + 1)) if training else [1] * (nc + 1)
    logit = logit + cin(p, E, nc, hp.get("cin_activation", "leaky_relu"), keep, masks.get("cin"))
    n = len(hp["deep_hidden_units"])
    keep = hp.get("deep_dropout", [1] * (n + 1)) if training else [1] * (n + 1)
    logit = logit + dnn(p, dnn_input(E, dense), n, hp.get("deep_activation", "leaky_relu"),
                        keep, masks.get("dnn"))
    return logit


def xdeepfm_l2(p, spec, hp):
    return (embedding_l2(p, spec, hp.get("embedding_l2_reg", 0.0))
            + linear_l2(p, hp.get("linear_l2_reg", 0.0))
            + dnn_l2(p, len(hp["deep_hidden_units"]), hp.get("deep_l2_reg", 0.0))
            + cin_l2(p, len(hp["cin_cross_layer_units"]), hp.get("cin_l2_reg", 0.0)))  # xDeepFM.py:106-114


MODELS = {
    "deepfm": (deepfm_logit, deepfm_l2),
    "dcn": (dcn_logit, dcn_l2),
    "xdeepfm": (xdeepfm_logit, xdeepfm_l2),
}


def model_loss(model, p, spec, idx, dense, y, hp, task="classification", masks=None,
               sparse_grad=False, mv=None):
    """_loss (xDeepFM.py:106-114): create_loss(y, _out(inputs)) + sum of layer l2().
    sparse_grad: table gradients as sparse tensors (bench.py's CPU baseline; needs all
    l2 factors 0, as TF's IndexedSlices stay sparse only then)."""
    global SPARSE_GRAD
    logit_fn, l2_fn = MODELS[model]
    SPARSE_GRAD = bool(sparse_grad)
    try:
        logit = logit_fn(p, spec, idx, dense, hp, True, masks, mv=mv)
    finally:
        SPARSE_GRAD = False
    pred = prediction(logit, task)
    loss = create_loss(y, pred, task)
    if not sparse_grad:
        loss = loss + l2_fn(p, spec, hp)
    return loss, logit, pred


def fwd_bwd(model, p, spec, idx, dense, y, hp, task="classification", masks=None, mv=None):
    """One forward+backward: returns (loss, logit [B], pred [B], grads dict).
    Embedding-table gradients come back dense (what TF's IndexedSlices + the
    dense l2 term add up to)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    loss, logit, pred = model_loss(model, leaves, spec, idx, dense, y, hp, task, masks, mv=mv)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return loss.detach(), logit.detach().reshape(-1), pred.detach(), grads
