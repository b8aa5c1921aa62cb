"""numpy restatement (forward only, any float dtype) of the reference layers.

TEST INFRASTRUCTURE - see oracle/__init__.py.  Literal on purpose: it forms the
one-hot matrices and the CIN outer-product tensor exactly the way the reference
does, so that it is an independent check of the (gather-form, chunked) torch
oracle in oracle/th_layers.py.  Use float64 for the high-precision twin.

Conventions shared with the torch oracle and the HIP path:
  idx    int64 [B, F]   per-field local row index, column order =
                        FeatureDictionary.embedding_feats (recman/tf/inputs.py:13-15)
  dense  float [B, Dn]  scaled dense features, column order = dense_feats
  tables list of F arrays [V_f, D]; bias tables list of F arrays [V_f, 1]
"""
import numpy as np


# --------------------------------------------------------------------------
# activations (TF semantics: tf.nn.leaky_relu default alpha = 0.2)
# --------------------------------------------------------------------------
def act_fn(name):
    if name in (None, "linear", "identity"):
        return lambda x: x
    if name == "relu":
        return lambda x: np.maximum(x, 0)
    if name == "leaky_relu":
        return lambda x: np.where(x > 0, x, 0.2 * x)
    if name == "sigmoid":
        return lambda x: 1.0 / (1.0 + np.exp(-x))
    if name == "tanh":
        return np.tanh
    raise ValueError(name)


def dropout(x, keep, mask=None):
    """tf.nn.dropout(x, rate=1-keep): identity at keep == 1, otherwise kept
    units are scaled by 1/keep (layers.py:461,466,589,602,707,740).  `mask` is
    the 0/1 keep mask, injected by the test (TF's RNG is not reproducible)."""
    if keep >= 1 or mask is None:
        return x
    return x * mask / keep


# --------------------------------------------------------------------------
# a6  FeatEmbedding / FeatEmbeddingLayer   (layers.py:117-128, 238-261)
# --------------------------------------------------------------------------
def feat_embedding_layer(idx, tables, bias_tables=None):
    """per field tf.nn.embedding_lookup(table_f, idx[:, f:f+1]) -> [B,1,D],
    concat over fields on axis 1 -> E [B,F,D]; same for the [V,1] bias tables
    -> [B,F,1] (None when use_bias=False)."""
    E = np.concatenate([tables[f][idx[:, f : f + 1]] for f in range(len(tables))], axis=1)
    bias = None
    if bias_tables is not None:
        bias = np.concatenate(
            [bias_tables[f][idx[:, f : f + 1]] for f in range(len(bias_tables))], axis=1
        )
    return E, bias


def embedding_l2(tables, l2_reg):
    """l2_reg * tf.nn.l2_loss(table) summed over fields, bias tables excluded
    (layers.py:188-193, 263-267).  l2_loss(x) = sum(x**2) / 2."""
    return sum(l2_reg * 0.5 * np.sum(np.square(t)) for t in tables)


# --------------------------------------------------------------------------
# a8  LinearCombiner + LinearLayer   (layers.py:270-354, 357-446; utils.py:51-67)
# --------------------------------------------------------------------------
def linear_one_hot_input(idx_cols, feat_sizes, dense, dtype):
    """LinearCombiner: one_hot(idx_f, depth=feat_size_f) per sparse feature then
    the dense columns, concatenated on axis 1 (layers.py:286-297).  Feature
    order is utils.py:31-36: sparse feats first, dense feats last."""
    B = idx_cols.shape[0]
    cols = []
    for f, V in enumerate(feat_sizes):
        oh = np.zeros((B, V), dtype=dtype)
        oh[np.arange(B), idx_cols[:, f]] = 1
        cols.append(oh)
    if dense is not None and dense.shape[1]:
        cols.append(dense.astype(dtype))
    return np.concatenate(cols, axis=1)


def linear_layer(x_onehot, W, W0, manual_weights=None):
    """tf.matmul(inputs, W) + W0 (layers.py:347, 439); with training=False the
    per-feature manual weights are added to W first (layers.py:338-345)."""
    if manual_weights is not None:
        W = W + manual_weights.reshape(-1, 1)
    return x_onehot @ W + W0


def linear_l2(W, l2_reg):
    return l2_reg * 0.5 * np.sum(np.square(W))  # layers.py:349-354


# --------------------------------------------------------------------------
# a9  FMLayer   (layers.py:457-478)
# --------------------------------------------------------------------------
def fm_layer(E, bias, keep=(1, 1), masks=(None, None)):
    bias = dropout(bias, keep[0], masks[0])
    y_first = np.sum(bias, axis=1)  # [B,1]                       :462
    E = dropout(E, keep[1], masks[1])
    sum_embeds = np.sum(E, axis=1, keepdims=True)  # [B,1,D]      :467
    square_of_sum = np.square(sum_embeds)  #                      :468
    sum_of_square = np.sum(np.square(E), axis=1, keepdims=True)  # :471-472
    y_second = 0.5 * (square_of_sum - sum_of_square)  #           :475
    y_second = np.sum(y_second, axis=2)  # [B,1]                  :476
    return y_first + y_second  #                                  :478


# --------------------------------------------------------------------------
# a10  DNNCombiner + DNN   (layers.py:494-501, 576-609)
# --------------------------------------------------------------------------
def dnn_combiner(E, dense):
    parts = [E.reshape(E.shape[0], -1)]
    if dense is not None and dense.shape[1]:
        parts.append(dense)
    return np.concatenate(parts, axis=1)


def dnn(x, weights, biases, w_out, w0_out, activation="relu", keep=None, masks=None):
    L = len(weights)
    keep = keep or [1] * (L + 1)
    masks = masks or [None] * (L + 1)
    y = dropout(x, keep[0], masks[0])  #                               :589
    act = act_fn(activation)
    for i in range(L):
        y = act(y @ weights[i] + biases[i])  #                         :593-601
        y = dropout(y, keep[i + 1], masks[i + 1])  #                   :602
    return y @ w_out + w0_out  #                                       :606-609


def dnn_l2(weights, w_out, l2_reg):
    return sum(l2_reg * 0.5 * np.sum(np.square(w)) for w in list(weights) + [w_out])  # :611-628


# --------------------------------------------------------------------------
# a11  CIN   (layers.py:697-760)
# --------------------------------------------------------------------------
def cin(E, filters, biases, cin_w, cin_w0, activation="leaky_relu", keep=None, masks=None,
        return_maps=False):
    """filters[i]: [1, m*H_i, N_i] (conv1d kernel, width 1).  Literal: splits the
    embedding axis, forms the D batched [B,m,1]x[B,1,H] outer products, reshapes
    to [D,B,m*H] (X0's field index is the major one), transposes to [B,D,m*H]
    and applies the filter as a matmul over the last axis."""
    B, m, D = E.shape
    L = len(filters)
    keep = keep or [1] * (L + 1)
    masks = masks or [None] * (L + 1)
    act = act_fn(activation)
    x0 = dropout(E, keep[0], masks[0])  #                                   :707
    hidden = [x0]
    finals = []
    field_nums = [m]
    maps = []
    split0 = [x0[:, :, d : d + 1] for d in range(D)]  # D x [B,m,1]         :711-713
    for i in range(L):
        size = filters[i].shape[-1]
        splitk = [hidden[-1][:, :, d : d + 1] for d in range(D)]  #         :715-719
        dot_m = np.stack(
            [split0[d] @ np.transpose(splitk[d], (0, 2, 1)) for d in range(D)], axis=0
        )  # [D,B,m,H]                                                      :721
        dot_o = dot_m.reshape(D, B, field_nums[0] * field_nums[i])  #       :722-725
        dot = np.transpose(dot_o, (1, 0, 2))  # [B,D,m*H]                   :726
        feat_map = dot @ filters[i][0]  # conv1d, kernel width 1, VALID     :728-733
        feat_map = feat_map + biases[i]  #                                  :734-736
        feat_map = act(feat_map)  #                                         :738
        feat_map = np.transpose(feat_map, (0, 2, 1))  # [B,N,D]             :739
        feat_map = dropout(feat_map, keep[i + 1], masks[i + 1])  #          :740
        maps.append(feat_map)
        field_nums.append(size // 2)  #                                     :742
        if i != L - 1:
            next_hidden = feat_map[:, : field_nums[-1]]  # tf.split 2*[size//2] :744-746
            direct = feat_map[:, field_nums[-1] : 2 * field_nums[-1]]
        else:
            direct = feat_map  #                                            :748
            next_hidden = None
        finals.append(direct)
        hidden.append(next_hidden)
    result = np.concatenate(finals, axis=1)  #                              :754
    pooled = np.sum(result, axis=-1)  # [B, final_size]                     :755
    logit = pooled @ cin_w + cin_w0  #                                      :757-760
    if return_maps:
        return logit, pooled, maps
    return logit


def cin_l2(filters, cin_w, l2_reg):
    return sum(l2_reg * 0.5 * np.sum(np.square(w)) for w in list(filters) + [cin_w])  # :762-777


# --------------------------------------------------------------------------
# a12  CrossNet - ABSENT from the reference (DCN.py:7, 134-137).
#      arXiv 1708.05123 eq. (3): x_{l+1} = x_0 (x_l^T w_l) + b_l + x_l
# --------------------------------------------------------------------------
def cross_net(x0, ws, bs, w_out):
    """ws, bs: [L, d]; w_out: [d, 1].  Returns logit [B,1] = x_L @ w_out
    (the call site needs a [B,1] logit that is added to dnn_logit, DCN.py:135-142)."""
    x = x0
    for l in range(ws.shape[0]):
        s = x @ ws[l].reshape(-1, 1)  # [B,1]
        x = x0 * s + bs[l] + x
    return x @ w_out


def cross_l2(ws, w_out, l2_reg):
    return l2_reg * 0.5 * (np.sum(np.square(ws)) + np.sum(np.square(w_out)))


# --------------------------------------------------------------------------
# a13 / a14  PredictionLayer + loss   (layers.py:796-808; utils.py:192-198)
# --------------------------------------------------------------------------
def prediction(logit, task="classification", global_bias=None):
    out = logit
    if global_bias is not None:
        out = out + global_bias
    if task == "classification":
        out = 1.0 / (1.0 + np.exp(-out))
    return out.reshape(-1)


KERAS_EPS = 1e-7


def binary_crossentropy(y_true, p):
    """tf.losses.binary_crossentropy on probabilities (Keras backend): clip p to
    [eps, 1-eps], bce = -(y log(p+eps) + (1-y) log(1-p+eps)), mean over the batch."""
    y = y_true.astype(p.dtype)
    pc = np.clip(p, KERAS_EPS, 1 - KERAS_EPS)
    bce = y * np.log(pc + KERAS_EPS) + (1 - y) * np.log(1 - pc + KERAS_EPS)
    return np.mean(-bce)


def mean_squared_error(y_true, p):
    return np.mean(np.square(p - y_true.astype(p.dtype)))
