"""Tensor-level wrappers over the C ABI: shape/dtype/device checks in Python (the
reference raises Python asserts before any arithmetic, layers.py:49,85,458,521-522),
raw pointers and sizes across the boundary, kernels enqueued on torch's current
stream.  torch is plumbing here: device memory and streams, nothing else.
"""
import ctypes

import torch

from . import _lib

F32, I64 = torch.float32, torch.int64


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, dtype, shape=None, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise ValueError(f"{name} must not be None")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (recman_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t.data_ptr()


def profile_marker(tag=0):
    """An empty marker kernel on the current stream (cuts a rocprofv3 trace, rm_profile_marker)."""
    _lib.call("rm_profile_marker", int(tag), _stream())


def embed_fwd(idx, table, field_off, *, bias_table=None, bias_ld=1, lin_w=None, lin_ld=1,
              lin_off=None, lin_w_dense=None, lin_w0=None, dense=None, mask_b=None, mask_e=None,
              E=None, fm_sum=None, fm_logit=None, lin_logit=None, table_ld=None, D=None,
              bias_col=None, lin_col=None, stream_rows=False):
    """Gather + FM + linear forward (see rm_embed_fwd in include/recman_hip.h).
    stream_rows: RM_EMBED_STREAM_ROWS - non-temporal row loads (ids with little reuse per batch).
    `table` is [R, table_ld]; D defaults to table.shape[1].  bias_col / lin_col: the FM
    bias / sparse linear weight live in that column of the table row itself (fused rows)."""
    B, F = idx.shape
    ld = table.shape[1] if table_ld is None else table_ld
    D = table.shape[1] if D is None else D
    Dn = 0 if dense is None else dense.shape[1]
    tp = _chk(table, "table", F32)
    fo = _chk(field_off, "field_off", I64, (F,))
    bp, bld = _chk(bias_table, "bias_table", F32, allow_none=True), bias_ld
    if bias_col is not None:
        bp, bld = tp + 4 * bias_col, ld
    lp, lld = _chk(lin_w, "lin_w", F32, allow_none=True), lin_ld
    lo = _chk(lin_off, "lin_off", I64, (F,), allow_none=True)
    if lin_col is not None:
        lp, lld, lo = tp + 4 * lin_col, ld, fo
    _lib.call(
        "rm_embed_fwd", _chk(idx, "idx", I64), tp, ld, fo, bp, bld, lp, lld, lo,
        _chk(lin_w_dense, "lin_w_dense", F32, (Dn,), allow_none=True),
        _chk(lin_w0, "lin_w0", F32, (1,), allow_none=True),
        _chk(dense, "dense", F32, (B, Dn), allow_none=True), Dn,
        _chk(mask_b, "mask_b", F32, (B, F), allow_none=True),
        _chk(mask_e, "mask_e", F32, (B, F, D), allow_none=True), B, F, D,
        _chk(E, "E", F32, (B, F, D), allow_none=True),
        _chk(fm_sum, "fm_sum", F32, (B, D), allow_none=True),
        _chk(fm_logit, "fm_logit", F32, (B,), allow_none=True),
        _chk(lin_logit, "lin_logit", F32, (B,), allow_none=True), 1 if stream_rows else 0, _stream())


def linear_fwd(idx, lin_off, w, dense, w_dense, w0, out):
    """out[b] = sum_f w[lin_off[f] + idx[b,f]] + dense[b,:] . w_dense + w0 (rm_linear_fwd)."""
    B = out.shape[0]
    F = 0 if idx is None else idx.shape[1]
    Dn = 0 if dense is None else dense.shape[1]
    _lib.call("rm_linear_fwd", _chk(idx, "idx", I64, (B, F), allow_none=True),
              _chk(lin_off, "lin_off", I64, (F,), allow_none=True), _chk(w, "w", F32, allow_none=True),
              _chk(dense, "dense", F32, (B, Dn), allow_none=True),
              _chk(w_dense, "w_dense", F32, (Dn,), allow_none=True),
              _chk(w0, "w0", F32, (1,), allow_none=True), B, F, Dn, _chk(out, "out", F32, (B,)), _stream())


def embed_bwd(d_rows, *, E=None, fm_sum=None, dE_up=None, g_fm=None, mask_b=None, mask_e=None,
              d_bias=None):
    B, F, D = d_rows.shape
    _lib.call(
        "rm_embed_bwd", _chk(E, "E", F32, (B, F, D), allow_none=True),
        _chk(fm_sum, "fm_sum", F32, (B, D), allow_none=True),
        _chk(dE_up, "dE_up", F32, (B, F, D), allow_none=True),
        _chk(g_fm, "g_fm", F32, (B,), allow_none=True),
        _chk(mask_b, "mask_b", F32, (B, F), allow_none=True),
        _chk(mask_e, "mask_e", F32, (B, F, D), allow_none=True), B, F, D,
        _chk(d_rows, "d_rows", F32), _chk(d_bias, "d_bias", F32, (B, F), allow_none=True),
        _stream())


def scatter_add_rows(d_table, idx, field_off, *, rows=None, g_row=None, width=None, ld=None):
    B, F = idx.shape
    if width is None:
        width = 1 if g_row is not None else rows.shape[-1]
    if ld is None:
        ld = d_table.shape[1] if d_table.dim() == 2 else 1
    _lib.call(
        "rm_scatter_add_rows", _chk(idx, "idx", I64), _chk(field_off, "field_off", I64, (F,)),
        _chk(rows, "rows", F32, allow_none=True), _chk(g_row, "g_row", F32, (B,), allow_none=True),
        B, F, width, ld, _chk(d_table, "d_table", F32), _stream())


def linear_dense_bwd(g, dense, d_w_dense, d_w0, workspace):
    B = g.shape[0]
    Dn = 0 if dense is None else dense.shape[1]
    if workspace.numel() < 256 * (Dn + 1):
        raise ValueError("linear_dense_bwd: workspace too small")
    if Dn > 1023:
        raise ValueError("linear_dense_bwd: at most 1023 columns")
    _lib.call(
        "rm_linear_dense_bwd", _chk(g, "g", F32, (B,)),
        _chk(dense, "dense", F32, (B, Dn), allow_none=True), B, Dn,
        _chk(d_w_dense, "d_w_dense", F32, (Dn,), allow_none=True),
        _chk(d_w0, "d_w0", F32, (1,), allow_none=True), _chk(workspace, "workspace", F32),
        _stream())


def logit_loss(branches, *, y=None, y_f=None, task="classification", logit=None, pred=None,
               dlogit=None, loss=None, workspace=None):
    """branches: up to four (tensor [B], coefficient) pairs."""
    if not 1 <= len(branches) <= 4:
        raise ValueError("logit_loss takes 1..4 branch logits")
    B = branches[0][0].shape[0]
    args = []
    for i in range(4):
        if i < len(branches):
            t, c = branches[i]
            args += [_chk(t, f"logit_{i}", F32, (B,)), float(c)]
        else:
            args += [None, 0.0]
    if loss is not None and (workspace is None or workspace.numel() < 1024):
        raise ValueError("logit_loss: loss needs a workspace of >= 1024 floats")
    _lib.call(
        "rm_logit_loss", *args, _chk(y, "y", I64, (B,), allow_none=True),
        _chk(y_f, "y_f", F32, (B,), allow_none=True), 0 if task == "classification" else 1, B,
        _chk(logit, "logit", F32, (B,), allow_none=True),
        _chk(pred, "pred", F32, (B,), allow_none=True),
        _chk(dlogit, "dlogit", F32, (B,), allow_none=True),
        _chk(loss, "loss", F32, (1,), allow_none=True),
        _chk(workspace, "workspace", F32, allow_none=True), _stream())


def cross_p_ld(L):
    """Row length of the saved dot products p [B, p_ld] (L+1 values, padded to a multiple of 4)."""
    return (L + 4) // 4 * 4


def cross_fwd(xe, xd, w, b, w_out, logit, p_out=None):
    """CrossNet forward (rm_cross_fwd).  p_out [B, cross_p_ld(L)]: the row's dot products
    (x0.w_l, x0.w_out), all the backward needs."""
    B, FD = xe.shape
    Dn = 0 if xd is None else xd.shape[1]
    L, d = w.shape
    if d != FD + Dn:
        raise ValueError(f"cross_fwd: w has d={d}, inputs have {FD}+{Dn}")
    p_ld = 0 if p_out is None else p_out.shape[1]
    _lib.call(
        "rm_cross_fwd", _chk(xe, "xe", F32), _chk(xd, "xd", F32, (B, Dn), allow_none=True), FD, Dn,
        _chk(w, "w", F32), _chk(b, "b", F32, (L, d)), _chk(w_out, "w_out", F32, (d,)), L, B,
        _chk(logit, "logit", F32, (B,)), _chk(p_out, "p_out", F32, (B, p_ld), allow_none=True), p_ld,
        _stream())


def cross_bwd(w, b, w_out, g, p, d_xe, coef, dx_in_e=None):
    """CrossNet backward (rm_cross_bwd): d_xe [B,FD] (+ dx_in_e), coef [B,2L+2]."""
    B, FD = d_xe.shape
    L, d = w.shape
    _lib.call(
        "rm_cross_bwd", FD, d - FD, _chk(w, "w", F32), _chk(b, "b", F32, (L, d)),
        _chk(w_out, "w_out", F32, (d,)), L, B, _chk(g, "g", F32, (B,)),
        _chk(p, "p", F32, (B, p.shape[1])), p.shape[1],
        _chk(dx_in_e, "dx_in_e", F32, (B, FD), allow_none=True), _chk(d_xe, "d_xe", F32, (B, FD)),
        _chk(coef, "coef", F32, (B, 2 * L + 2)), _stream())


def cross_param_grads(P, colsum, w, b, w_out, d_w, d_b, d_w_out):
    L, d = w.shape
    _lib.call(
        "rm_cross_param_grads", _chk(P, "P", F32, (d, L + 1)),
        _chk(colsum, "colsum", F32, (L + 1,)), _chk(w, "w", F32), _chk(b, "b", F32, (L, d)),
        _chk(w_out, "w_out", F32, (d,)), L, d, _chk(d_w, "d_w", F32, (L, d)),
        _chk(d_b, "d_b", F32, (L, d)), _chk(d_w_out, "d_w_out", F32, (d,)), _stream())


def gather_rows(table, rows, out):
    """out[i, :] = table[rows[i], :width] with width = out.shape[1] <= table.shape[1] (the shard keeps
    optimizer state behind the exchanged columns).  `table` may live in PINNED host memory (th/feeder.py)."""
    n = rows.shape[0]
    width = out.shape[1]
    tp, ld, cols = _rows2d(table, "table", allow_pinned=True)  # (a [:, :width] view of a wider shard is fine)
    if cols < width:
        raise ValueError(f"gather_rows: table has {cols} columns, out needs {width}")
    _lib.call("rm_gather_rows", tp, ld, _chk(rows, "rows", I64, (n,)), n, width,
              _chk(out, "out", F32, (n, width)), _stream())


def permute_rows(src, slot, dst, inverse=False):
    n, width = src.shape
    _lib.call("rm_permute_rows", _chk(src, "src", F32), _chk(slot, "slot", I64, (n,)), n, width,
              1 if inverse else 0, _chk(dst, "dst", F32, (n, width)), _stream())


ACT_IDS = {"identity": 0, "relu": 1, "leaky_relu": 2}


def cin_filter_workspace(m, H, N):
    return int(_lib.lib().rm_cin_filter_workspace(m, H, N))


def cin_filter_workspace6(m, H, N, D):
    """Floats of filter workspace for the bf16x6 form of cin_layer_fwd (0: the shape is not covered)."""
    return int(_lib.lib().rm_cin_filter_workspace6(int(m), int(H), int(N), int(D)))


def cin_layer_fwd(X0, Xk, H, W, bias, act, out, filter_ws, pooled=None, pool_col0=0, pool_from=0, ws6=None,
                  first6=False):
    """One CIN layer forward.  X0 [B,m,D]; Xk [B,Hk,D] of which rows j < H are used;
    W [m*H, N]; out [B,N,D]; pooled [B, P] gets sum_d out[:, pool_from:, :] at pool_col0.
    ws6 (cin_filter_workspace6 floats): on the bf16 matrix pipe with split operands (rm_cin_layer_fwd6) when that
    kernel covers the layer; returns True when it ran."""
    B, m, D = X0.shape
    N = W.shape[1]
    if ws6 is not None and (first6 or Xk.data_ptr() != X0.data_ptr()):
        need = cin_filter_workspace6(m, H, N, D)
        if need > 0 and ws6.numel() >= need:
            if W.shape[0] != m * H or Xk.shape[0] != B or Xk.shape[2] != D or Xk.shape[1] < H:
                raise ValueError("cin_layer_fwd: shape mismatch")
            _lib.call(
                "rm_cin_layer_fwd6", _chk(X0, "X0", F32), _chk(Xk, "Xk", F32), Xk.shape[1] * D,
                _chk(W, "W", F32), _chk(bias, "bias", F32, (N,)), ACT_IDS[act], B, m, H, N, D,
                _chk(out, "out", F32, (B, N, D)), _chk(pooled, "pooled", F32, allow_none=True),
                0 if pooled is None else pooled.shape[1], pool_col0, pool_from,
                _chk(ws6, "ws6", F32), _stream())
            return True
    if W.shape[0] != m * H:
        raise ValueError(f"cin_layer_fwd: filter has {W.shape[0]} rows, expected m*H = {m * H}")
    if Xk.shape[0] != B or Xk.shape[2] != D or Xk.shape[1] < H:
        raise ValueError("cin_layer_fwd: Xk shape mismatch")
    if filter_ws.numel() < cin_filter_workspace(m, H, N):
        raise ValueError("cin_layer_fwd: filter workspace too small")
    _lib.call(
        "rm_cin_layer_fwd", _chk(X0, "X0", F32), _chk(Xk, "Xk", F32), Xk.shape[1] * D,
        _chk(W, "W", F32), _chk(bias, "bias", F32, (N,)), ACT_IDS[act], B, m, H, N, D,
        _chk(out, "out", F32, (B, N, D)), _chk(pooled, "pooled", F32, allow_none=True),
        0 if pooled is None else pooled.shape[1], pool_col0, pool_from,
        _chk(filter_ws, "filter_ws", F32), _stream())


def cin_bwd_workspace(B, m, H, N, D):
    return int(_lib.lib().rm_cin_bwd_workspace(B, m, H, N, D))


def cin_layer_bwd(X0, Xk, H, W, act, out, g, dX0, dW, dbias, workspace, *, xk_is_x0=False,
                  d_hidden=None, cin_w_direct=None, pool_from=0, accumulate_dx0=True, dXk=None, split=False,
                  first6=False):
    """One CIN layer backward (see rm_cin_layer_bwd).  d_hidden [B,pool_from,D] is the next
    layer's dXk; cin_w_direct [N-pool_from] the cin_w entries of this layer's direct half.  split: the dX pass on
    the bf16 matrix pipe with split fp32 operands where csrc/cin6.hip covers the layer."""
    B, m, D = X0.shape
    N = W.shape[1]
    if workspace.numel() < cin_bwd_workspace(B, m, H, N, D):
        raise ValueError("cin_layer_bwd: workspace too small")
    _lib.call(
        "rm_cin_layer_bwd", _chk(X0, "X0", F32), _chk(Xk, "Xk", F32), Xk.shape[1] * D,
        1 if xk_is_x0 else 0, _chk(W, "W", F32, (m * H, N)), ACT_IDS[act],
        _chk(out, "out", F32, (B, N, D)), _chk(d_hidden, "d_hidden", F32, allow_none=True),
        0 if d_hidden is None else d_hidden.shape[1] * D, _chk(g, "g", F32, (B,)),
        _chk(cin_w_direct, "cin_w_direct", F32, (N - pool_from,), allow_none=True), pool_from,
        B, m, H, N, D, _chk(dX0, "dX0", F32, (B, m, D)), (1 if accumulate_dx0 else 0) | (2 if split else 0) | (4 if first6 else 0),
        _chk(dXk, "dXk", F32, allow_none=True), 0 if dXk is None else dXk.shape[1] * D,
        _chk(dW, "dW", F32, (m * H, N)), _chk(dbias, "dbias", F32, (N,)),
        _chk(workspace, "workspace", F32), workspace.numel(), _stream())


def rowdot(X, w, w0, out):
    """out[b] = X[b,:] . w + w0 (the [*,1] output projections)."""
    B, P_ = X.shape
    _lib.call("rm_rowdot", _chk(X, "X", F32), _chk(w, "w", F32, (P_,)),
              _chk(w0, "w0", F32, (1,), allow_none=True), B, P_, _chk(out, "out", F32, (B,)),
              _stream())


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _int_array(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def mlp_supported(FD, Dn, hidden):
    return bool(_lib.lib().rm_mlp_supported(FD, Dn, len(hidden), _int_array(hidden)))


def mlp_bwd_workspace(FD, Dn):
    return int(_lib.lib().rm_mlp_bwd_workspace(FD, Dn))


def mlp_tail(B, branches, coef_mlp, *, y=None, y_f=None, task="classification", grad_scale=1.0,
             logit=None, pred=None, dlogit, loss_partial, loss=None, dh):
    """Builds the rm_mlp_tail struct of the fused training head (see include/recman_hip.h):
    branches = up to two (tensor [B], coefficient) pairs summed BEFORE the MLP's own logit.
    Returns the struct; pass it to mlp_fwd and then to mlp_bwd (the tensors must stay alive)."""
    if len(branches) > 2:
        raise ValueError("mlp_tail takes at most two other branch logits")
    if (y is None) == (y_f is None):
        raise ValueError("mlp_tail needs exactly one of y / y_f")
    if loss_partial.numel() < (B + 31) // 32:
        raise ValueError("mlp_tail: loss_partial needs ceil(B/32) floats")
    t = _lib.MlpTail()
    for name, i in (("a", 0), ("b", 1)):
        if i < len(branches):
            setattr(t, f"logit_{name}", _chk(branches[i][0], f"logit_{name}", F32, (B,)))
            setattr(t, f"coef_{name}", float(branches[i][1]))
    t.coef_mlp = float(coef_mlp)
    t.y = _chk(y, "y", I64, (B,), allow_none=True)
    t.y_f = _chk(y_f, "y_f", F32, (B,), allow_none=True)
    t.task = 0 if task == "classification" else 1
    t.grad_scale = float(grad_scale)
    t.logit = _chk(logit, "logit", F32, (B,), allow_none=True)
    t.pred = _chk(pred, "pred", F32, (B,), allow_none=True)
    t.dlogit = _chk(dlogit, "dlogit", F32, (B,))
    t.loss_partial = _chk(loss_partial, "loss_partial", F32)
    t.loss = _chk(loss, "loss", F32, (1,), allow_none=True)
    for l, d in enumerate(dh):
        t.dh[l] = _chk(d, f"dh[{l}]", F32, (B, 32))
    return t


def _tail_ref(tail):
    import ctypes

    return None if tail is None else ctypes.cast(ctypes.pointer(tail), ctypes.c_void_p)


def mlp_fwd(xe, xd, Ws, bs, w_out, w0_out, act, h_out, logit, tail=None):
    """Fused skinny-MLP forward.  Ws[l] / bs[l]: layer weights; h_out[l] [B,32].
    tail (mlp_tail(...)): also the final logit, prediction, loss terms, dLoss/dlogit and the dh chain."""
    B, FD = xe.shape
    Dn = 0 if xd is None else xd.shape[1]
    H = [W.shape[1] for W in Ws]
    for l, W in enumerate(Ws):
        _chk(W, f"W[{l}]", F32, (FD + Dn if l == 0 else H[l - 1], H[l]))
        _chk(bs[l], f"bias[{l}]", F32, (H[l],))
        _chk(h_out[l], f"h_out[{l}]", F32, (B, 32))
    _lib.call("rm_mlp_fwd", _chk(xe, "xe", F32), _chk(xd, "xd", F32, (B, Dn), allow_none=True), FD,
              Dn, len(Ws), _int_array(H), _ptr_array(Ws), _ptr_array(bs),
              _chk(w_out, "w_out", F32, (H[-1],)), _chk(w0_out, "w0_out", F32, (1,)), ACT_IDS[act], B,
              _ptr_array(h_out), _chk(logit, "logit", F32, (B,)), _tail_ref(tail), _stream())


def embed_mlp_fwd_supported(F, D, table_ld, Dn, hidden):
    """rm_embed_mlp_fwd_supported: the one-kernel gather + FM + linear + MLP forward covers this shape."""
    return bool(_lib.lib().rm_embed_mlp_fwd_supported(int(F), int(D), int(table_ld), int(Dn), len(hidden),
                                                      _int_array(list(hidden))))


def embed_mlp_fwd(idx, rows, field_off, D, table_ld, xd, Ws, bs, w_out, w0_out, act, E, h_out, logit,
                  want_bias=False, want_lin=False, lin_w_dense=None, lin_w0=None, fm_sum=None, fm_logit=None,
                  lin_logit=None, stream_rows=False, tail=None):
    """rm_embed_mlp_fwd: rm_embed_fwd (fused rows [D | bias | lin | ...]) + rm_mlp_fwd in one kernel."""
    B, F = idx.shape
    Dn = 0 if xd is None else xd.shape[1]
    H = [W.shape[1] for W in Ws]
    for l, W in enumerate(Ws):
        _chk(W, f"W[{l}]", F32, (F * D + Dn if l == 0 else H[l - 1], H[l]))
        _chk(bs[l], f"bias[{l}]", F32, (H[l],))
        _chk(h_out[l], f"h_out[{l}]", F32, (B, 32))
    if rows.dim() != 2 or rows.shape[1] != table_ld or not rows.is_contiguous():
        raise ValueError("embed_mlp_fwd: rows must be the contiguous fused table [R, table_ld]")
    _lib.call("rm_embed_mlp_fwd", _chk(idx, "idx", I64), _chk(rows, "rows", F32), int(table_ld),
              _chk(field_off, "field_off", I64, (F,)), int(bool(want_bias)), int(bool(want_lin)),
              _chk(lin_w_dense, "lin_w_dense", F32, (Dn,), allow_none=True),
              _chk(lin_w0, "lin_w0", F32, (1,), allow_none=True), _chk(xd, "xd", F32, (B, Dn), allow_none=True),
              Dn, B, F, int(D), _chk(E, "E", F32), _chk(fm_sum, "fm_sum", F32, (B, D), allow_none=True),
              _chk(fm_logit, "fm_logit", F32, (B,), allow_none=True),
              _chk(lin_logit, "lin_logit", F32, (B,), allow_none=True), 1 if stream_rows else 0, len(Ws),
              _int_array(H), _ptr_array(Ws), _ptr_array(bs), _chk(w_out, "w_out", F32, (H[-1],)),
              _chk(w0_out, "w0_out", F32, (1,)), ACT_IDS[act], _ptr_array(h_out), _chk(logit, "logit", F32, (B,)),
              _tail_ref(tail), _stream())


def mlp_bwd(xe, xd, Ws, w_out, act, g, h, d_rows, dh, dW, workspace, fm_sum=None, db=None,
            d_w_out=None, d_w0_out=None, d_xd_wsum=None, d_g_sum=None, tail=None, stream_d_rows=False):
    """tail: the struct the forward ran with - dh is already there (no chain launch) and the
    finishing kernel also reduces the loss.  stream_d_rows: RM_MLP_STREAM_DROWS (non-temporal d_rows
    stores: only when no optimizer step re-reads them)."""
    B, FD = xe.shape
    Dn = 0 if xd is None else xd.shape[1]
    H = [W.shape[1] for W in Ws]
    for l, W in enumerate(Ws):
        _chk(W, f"W[{l}]", F32, (FD + Dn if l == 0 else H[l - 1], H[l]))
        _chk(dW[l], f"dW[{l}]", F32, tuple(W.shape))
        _chk(h[l], f"h[{l}]", F32, (B, 32))
        _chk(dh[l], f"dh[{l}]", F32, (B, 32))
    if workspace.numel() < mlp_bwd_workspace(FD, Dn):
        raise ValueError("mlp_bwd: workspace too small")
    D = 0 if fm_sum is None else fm_sum.shape[1]
    _lib.call("rm_mlp_bwd", _chk(xe, "xe", F32), _chk(xd, "xd", F32, (B, Dn), allow_none=True), FD,
              Dn, len(Ws), _int_array(H), _ptr_array(Ws), _chk(w_out, "w_out", F32, (H[-1],)),
              ACT_IDS[act], B, _chk(g, "g", F32, (B,)), _ptr_array(h),
              _chk(fm_sum, "fm_sum", F32, allow_none=True), D, _chk(d_rows, "d_rows", F32, (B, FD)),
              _ptr_array(dh), _ptr_array(dW), None if db is None else _ptr_array(db),
              _chk(d_w_out, "d_w_out", F32, (H[-1],), allow_none=True),
              _chk(d_w0_out, "d_w0_out", F32, (1,), allow_none=True),
              _chk(d_xd_wsum, "d_xd_wsum", F32, (Dn,), allow_none=True),
              _chk(d_g_sum, "d_g_sum", F32, (1,), allow_none=True),
              _chk(workspace, "workspace", F32), _tail_ref(tail), 1 if stream_d_rows else 0, _stream())


def deepfm_step_supported(F, D, table_ld, Dn, hidden):
    """rm_deepfm_step_supported: the one-kernel DeepFM training step covers this shape."""
    return bool(_lib.lib().rm_deepfm_step_supported(int(F), int(D), int(table_ld), int(Dn), len(hidden),
                                                    _int_array(list(hidden))))


def deepfm_step_workspace(F, Dn):
    return int(_lib.lib().rm_deepfm_step_workspace(int(F), int(Dn)))


def deepfm_step(idx, rows, field_off, D, table_ld, dense, y, Ws, bs, w_out, w0_out, lin_w_dense, lin_w0, act, task,
                d_rows, logit, pred, dlogit, loss, dW, db, d_w_out, d_w0_out, d_lin_w_dense, d_lin_w0, workspace,
                grad_scale=1.0, stream_rows=False, stream_d_rows=False, skip_finish=False, packed_rows=0,
                lin_field_mask=None):
    """rm_deepfm_step: DeepFM's forward + every gradient in one kernel (+ the finishing reduction; skip_finish =
    measurement only, the parameter gradients and the loss are then not written).  packed_rows > 0: the row-sharded
    form - rows = the received rows [packed_rows, D + 4], idx = positions in it, d_rows = the gradient send buffer
    [packed_rows, D + 4] written in bucketed order (see include/recman_hip.h)."""
    B, F = idx.shape
    Dn = 0 if dense is None else dense.shape[1]
    H = [W.shape[1] for W in Ws]
    for l, W in enumerate(Ws):
        _chk(W, f"W[{l}]", F32, (F * D + Dn if l == 0 else H[l - 1], H[l]))
        _chk(bs[l], f"bias[{l}]", F32, (H[l],))
        _chk(dW[l], f"dW[{l}]", F32, tuple(W.shape))
        _chk(db[l], f"db[{l}]", F32, (H[l],))
    if rows.dim() != 2 or rows.shape[1] != table_ld or not rows.is_contiguous():
        raise ValueError("deepfm_step: rows must be the contiguous fused table [R, table_ld]")
    if packed_rows and (d_rows.numel() < packed_rows * (D + 4) or rows.shape[0] < packed_rows or table_ld != D + 4):
        raise ValueError("deepfm_step: packed form needs rows / d_rows of [packed_rows, D + 4]")
    if rows.shape[0] >= 1 << 32:
        raise ValueError("deepfm_step: the table has more than 2^32 rows")
    if workspace.numel() < deepfm_step_workspace(F, Dn):
        raise ValueError("deepfm_step: workspace too small")
    yk = (_chk(y, "y", I64, (B,)), None) if y.dtype == I64 else (None, _chk(y, "y_f", F32, (B,)))
    _lib.call("rm_deepfm_step", _chk(idx, "idx", I64), _chk(rows, "rows", F32), int(table_ld),
              _chk(field_off, "field_off", I64, (F,)), _chk(dense, "dense", F32, (B, Dn), allow_none=True), Dn,
              yk[0], yk[1], B, F, int(D), len(Ws), _int_array(H), _ptr_array(Ws), _ptr_array(bs),
              _chk(w_out, "w_out", F32, (H[-1],)), _chk(w0_out, "w0_out", F32, (1,)),
              _chk(lin_w_dense, "lin_w_dense", F32, (Dn,), allow_none=True), _chk(lin_w0, "lin_w0", F32, (1,)),
              ACT_IDS[act], 0 if task == "classification" else 1, float(grad_scale),
              _chk(d_rows, "d_rows", F32), _chk(logit, "logit", F32, (B,)), _chk(pred, "pred", F32, (B,)),
              _chk(dlogit, "dlogit", F32, (B,)), _chk(loss, "loss", F32, (1,)), _ptr_array(dW), _ptr_array(db),
              _chk(d_w_out, "d_w_out", F32, (H[-1],)), _chk(d_w0_out, "d_w0_out", F32, (1,)),
              _chk(d_lin_w_dense, "d_lin_w_dense", F32, (Dn,), allow_none=True),
              _chk(d_lin_w0, "d_lin_w0", F32, (1,)), _chk(workspace, "workspace", F32), int(packed_rows),
              _chk(lin_field_mask, "lin_field_mask", F32, (F,), allow_none=True),
              (1 if stream_rows else 0) | (2 if stream_d_rows else 0) | (4 if skip_finish else 0), _stream())


def shard_route(idx, field_off, world, pos, send_ids, counts, workspace):
    B, F = idx.shape
    n = B * F
    need = int(_lib.lib().rm_shard_route_workspace(world))
    if workspace.dtype != torch.int32 or workspace.numel() < need:
        raise ValueError(f"shard_route: workspace must be int32 with >= {need} elements")
    _lib.call("rm_shard_route", _chk(idx, "idx", I64), _chk(field_off, "field_off", I64, (F,)), B, F,
              world, _chk(pos, "pos", I64, (n,)), _chk(send_ids, "send_ids", I64, (n,)),
              _chk(counts, "counts", I64, (world,)), workspace.data_ptr(), _stream())


def shard_route_padded(idx, field_off, world, cap, pos, send_ids, counts, overflow, workspace):
    B, F = idx.shape
    n = B * F
    need = int(_lib.lib().rm_shard_route_workspace(world))
    if workspace.dtype != torch.int32 or workspace.numel() < need:
        raise ValueError(f"shard_route_padded: workspace must be int32 with >= {need} elements")
    if overflow.dtype != torch.int32 or overflow.numel() < 1:
        raise ValueError("shard_route_padded: overflow must be an int32 flag")
    _lib.call("rm_shard_route_padded", _chk(idx, "idx", I64), _chk(field_off, "field_off", I64, (F,)), B, F,
              world, int(cap), _chk(pos, "pos", I64, (n,)), _chk(send_ids, "send_ids", I64, (world * cap,)),
              _chk(counts, "counts", I64, (world,)), overflow.data_ptr(), workspace.data_ptr(), _stream())


def pack_grad_rows(d_rows, g_bias, g_lin, pos, out, lin_field_mask=None):
    B, F, D = d_rows.shape
    n, width = out.shape
    if n < B * F:
        raise ValueError("pack_grad_rows: out must have at least B*F rows")
    _lib.call("rm_pack_grad_rows", _chk(d_rows, "d_rows", F32),
              _chk(g_bias, "g_bias", F32, (B,), allow_none=True),
              _chk(g_lin, "g_lin", F32, (B,), allow_none=True),
              _chk(lin_field_mask, "lin_field_mask", F32, (F,), allow_none=True),
              _chk(pos, "pos", I64, (B * F,)), B, F, D, width, _chk(out, "out", F32), _stream())


OPT_KINDS = {"adam": 0, "adagrad": 1, "gd": 2, "sgd": 2}


def sparse_optimizer_workspace(n):
    """Bytes of workspace rm_sparse_optimizer_step needs for n occurrences."""
    need = int(_lib.lib().rm_sparse_optimizer_workspace(int(n)))
    if need < 0:
        raise ValueError(f"sparse_optimizer_workspace: bad occurrence count {n}")
    return need


def _opt_ws(workspace, n):
    if workspace.dtype != torch.uint8 or not workspace.is_cuda or workspace.numel() < sparse_optimizer_workspace(n):
        raise ValueError("sparse optimizer: workspace must be a uint8 device tensor of "
                         "sparse_optimizer_workspace(n) bytes")
    return workspace.data_ptr(), workspace.numel()


def sparse_optimizer_prepare(workspace, R, idx=None, field_off=None, row_ids=None, max_field_rows=0):
    """The id-only part of a row-wise step (keys + stable sort) on the current stream
    (rm_sparse_optimizer_prepare); follow it with sparse_optimizer_step(..., prepared=True).
    max_field_rows > 0: the fields own disjoint ascending row ranges of at most that many rows (the sort then runs
    per field on the local ids)."""
    if row_ids is not None:
        n, F = row_ids.numel(), 1
    else:
        n, F = idx.numel(), idx.shape[1]
    wp, wn = _opt_ws(workspace, n)
    _lib.call("rm_sparse_optimizer_prepare", _chk(idx, "idx", I64, allow_none=True),
              _chk(field_off, "field_off", I64, allow_none=True), _chk(row_ids, "row_ids", I64, allow_none=True),
              n, F, int(R), int(max_field_rows), wp, wn, _stream())


def sparse_optimizer_step(idx, field_off, d_rows, rows, mom, workspace, step, kind, lr, D=None,
                          g_bias=None, g_lin=None, reset=False, beta1=0.9, beta2=0.999, eps=1e-7,
                          lin_field_mask=None, prepared=False, l2_embedding=0.0, l2_linear=0.0, max_field_rows=0):
    """Lazy row-wise optimizer step on table rows [R, ld] (see rm_sparse_optimizer_step): rows =
    [D emb | bias | lin | m_b | m_l | v_b | v_l | pad], mom [R, 2D] = [m | v] of the embedding."""
    B, F, Dg = d_rows.shape
    D = Dg if D is None else D
    R, ld = rows.shape
    if mom is not None:
        _chk(mom, "mom", F32, (R, 2 * D))
    wp, wn = _opt_ws(workspace, B * F)
    _lib.call("rm_sparse_optimizer_step", _chk(idx, "idx", I64, (B, F)),
              _chk(field_off, "field_off", I64, (F,)), _chk(d_rows, "d_rows", F32, (B, F, D)),
              _chk(g_bias, "g_bias", F32, (B,), allow_none=True),
              _chk(g_lin, "g_lin", F32, (B,), allow_none=True), B, F, D, R, _chk(rows, "rows", F32), ld,
              None if mom is None else mom.data_ptr(), int(step), OPT_KINDS[kind], float(lr), float(beta1),
              float(beta2), float(eps), 1 if reset else 0, float(l2_embedding), float(l2_linear),
              _chk(lin_field_mask, "lin_field_mask", F32, (F,), allow_none=True), int(max_field_rows),
              1 if prepared else 0, wp, wn, _stream())


def sparse_optimizer_step_rows(row_ids, grad_rows, D, rows, mom, workspace, step, kind, lr, reset=False,
                               beta1=0.9, beta2=0.999, eps=1e-7, prepared=False, l2_embedding=0.0, l2_linear=0.0):
    """The same step from gradient rows that carry their (local) table row: row_ids [n] (< 0: skip),
    grad_rows [n, gw] = [dE | g_bias | g_lin | ...] (rm_sparse_optimizer_step_rows)."""
    n, gw = grad_rows.shape
    R, ld = rows.shape
    if mom is not None:
        _chk(mom, "mom", F32, (R, 2 * D))
    wp, wn = _opt_ws(workspace, n)
    _lib.call("rm_sparse_optimizer_step_rows", _chk(row_ids, "row_ids", I64, (n,)),
              _chk(grad_rows, "grad_rows", F32), gw, n, D, R, _chk(rows, "rows", F32), ld,
              None if mom is None else mom.data_ptr(), int(step), OPT_KINDS[kind], float(lr), float(beta1),
              float(beta2), float(eps), 1 if reset else 0, float(l2_embedding), float(l2_linear),
              1 if prepared else 0, wp, wn, _stream())


def dense_optimizer_step(p, g, m, v, step, kind, lr, reset=False, beta1=0.9, beta2=0.999, eps=1e-7):
    """One launch over a flat parameter buffer (rm_dense_optimizer_step)."""
    n = p.numel()
    _lib.call("rm_dense_optimizer_step", _chk(p, "p", F32, (n,)), _chk(g, "g", F32, (n,)),
              _chk(m, "m", F32, (n,), allow_none=True), _chk(v, "v", F32, (n,), allow_none=True), n,
              int(step), OPT_KINDS[kind], float(lr), float(beta1), float(beta2), float(eps),
              1 if reset else 0, _stream())


def _chk_csr(offsets, ids, vals):
    if int(offsets.shape[0]) < 1:
        raise ValueError("CSR offsets must have B+1 entries")
    if vals is not None and vals.shape != ids.shape:
        raise ValueError(f"vals {tuple(vals.shape)} must match ids {tuple(ids.shape)}")


def pool_rows(rows, row0, D, offsets, ids, out, vals=None):
    """Pooled fused rows of a multi-valued feature (rm_pool_rows): out [B, LD].  vals=None:
    sqrtn combiner (MultiValCsvFeat); vals [nnz]: value-weighted (SparseValueFeat)."""
    B = offsets.shape[0] - 1
    LD = rows.shape[1]
    _chk_csr(offsets, ids, vals)
    _lib.call("rm_pool_rows", _chk(rows, "rows", F32), int(row0), LD, D, _chk(offsets, "offsets", I64),
              _chk(ids, "ids", I64), _chk(vals, "vals", F32, allow_none=True), B,
              _chk(out, "out", F32, (B, LD)), _stream())


def pool_rows_bwd(d_rows_f, g_bias, g_lin, D, offsets, ids, row0, d_table, d_bias, d_lin, vals=None):
    """d_rows_f: [B, D] view (row stride may be larger) of the pooled rows' gradient."""
    B = offsets.shape[0] - 1
    _chk_csr(offsets, ids, vals)
    if d_rows_f.stride(1) != 1:
        raise ValueError("pool_rows_bwd: d_rows_f must be unit-stride along D")
    _lib.call("rm_pool_rows_bwd", d_rows_f.data_ptr(), d_rows_f.stride(0),
              _chk(g_bias, "g_bias", F32, (B,), allow_none=True),
              _chk(g_lin, "g_lin", F32, (B,), allow_none=True), D, _chk(offsets, "offsets", I64),
              _chk(ids, "ids", I64), _chk(vals, "vals", F32, allow_none=True), B, int(row0),
              _chk(d_table, "d_table", F32),
              _chk(d_bias, "d_bias", F32, allow_none=True), _chk(d_lin, "d_lin", F32, allow_none=True),
              _stream())


def _cols(t, name, dtype, B, T):
    """(data pointer, row stride) of a [B, T] view with unit-stride columns (a column block of a wider matrix)."""
    if t is None:
        return None, 0
    if not t.is_cuda or t.dtype != dtype or tuple(t.shape) != (B, T) or (T > 1 and t.stride(1) != 1):
        raise ValueError(f"{name}: expected a cuda {dtype} [B={B}, T={T}] view with unit-stride columns, got "
                         f"{t.dtype} {tuple(t.shape)} strides {t.stride()} on {t.device}")
    return t.data_ptr(), t.stride(0)


def pool_rows_padded(rows, D, pos, ids, out, vals=None):
    """rm_pool_rows_padded: pooled rows out [B, LD] from the tag rows rows[pos[b, t]] (ids[b, t] < 0: no tag);
    pos / ids / vals are [B, T] views (column blocks of the occurrence matrix)."""
    B, T = ids.shape
    LD = rows.shape[1]
    pp, pl = _cols(pos, "pos", I64, B, T)
    ip, il = _cols(ids, "ids", I64, B, T)
    vp, vl = _cols(vals, "vals", F32, B, T)
    _lib.call("rm_pool_rows_padded", _chk(rows, "rows", F32), LD, D, pp, pl, ip, il, vp, vl, B, T,
              _chk(out, "out", F32, (B, LD)), _stream())


def pack_pooled_grad_rows(d_rows_f, g_bias, g_lin, D, pos, ids, out, vals=None):
    """rm_pack_pooled_grad_rows: the tags' gradient rows written at out[pos[b, t]] (out [slots, width]);
    d_rows_f: [B, D] view (row stride may be larger) of the pooled rows' gradient."""
    B, T = ids.shape
    if d_rows_f.stride(1) != 1:
        raise ValueError("pack_pooled_grad_rows: d_rows_f must be unit-stride along D")
    pp, pl = _cols(pos, "pos", I64, B, T)
    ip, il = _cols(ids, "ids", I64, B, T)
    vp, vl = _cols(vals, "vals", F32, B, T)
    _lib.call("rm_pack_pooled_grad_rows", d_rows_f.data_ptr(), d_rows_f.stride(0),
              _chk(g_bias, "g_bias", F32, (B,), allow_none=True), _chk(g_lin, "g_lin", F32, (B,), allow_none=True),
              D, pp, pl, ip, il, vp, vl, B, T, out.shape[1], _chk(out, "out", F32), _stream())


def bias_act_(x, bias, act):
    B, N = x.shape
    _lib.call("rm_bias_act", _chk(x, "x", F32), _chk(bias, "bias", F32, (N,), allow_none=True), B, N,
              ACT_IDS[act], _stream())


def outer_actgrad(g, w, a, act, da):
    """da[b,j] = g[b] * w[j] * act'(a[b,j]) (a may be None: no activation factor)."""
    B, N = da.shape
    _lib.call("rm_outer_actgrad", _chk(g, "g", F32, (B,)), _chk(w, "w", F32, (N,)),
              _chk(a, "a", F32, (B, N), allow_none=True), B, N, ACT_IDS[act], _chk(da, "da", F32), _stream())


def outer_actgrad_sums_workspace(B, N):
    return int(_lib.lib().rm_outer_actgrad_sums_workspace(B, N))


def outer_actgrad_sums(g, w, a, act, da, d_w, d_w0, db, workspace):
    """outer_actgrad + d_w[j] = sum_b g[b] a[b,j], d_w0 = sum_b g[b], db[j] = sum_b da[b,j] in one pass."""
    B, N = da.shape
    if workspace.numel() < outer_actgrad_sums_workspace(B, N):
        raise ValueError("outer_actgrad_sums: workspace too small")
    _lib.call("rm_outer_actgrad_sums", _chk(g, "g", F32, (B,)), _chk(w, "w", F32, (N,)),
              _chk(a, "a", F32, (B, N)), B, N, ACT_IDS[act], _chk(da, "da", F32),
              _chk(d_w, "d_w", F32, (N,), allow_none=True), _chk(d_w0, "d_w0", F32, (1,), allow_none=True),
              _chk(db, "db", F32, (N,), allow_none=True), _chk(workspace, "workspace", F32), _stream())


def act_bwd_(da, a, act):
    B, N = da.shape
    _lib.call("rm_act_bwd", _chk(da, "da", F32), _chk(a, "a", F32, (B, N)), B, N, ACT_IDS[act], _stream())


# ---- wide dense layers (csrc/gemm.hip) -------------------------------------------------------
DENSE_BIAS_ACT, DENSE_MUL_ACTGRAD, DENSE_ADD, DENSE_CROSS = 0, 1, 2, 3


def _rows2d(t, name, allow_none=False, allow_pinned=False):
    """A 2-D f32 tensor with unit column stride -> (pointer, leading dimension, columns).  allow_pinned: a
    PINNED host tensor is accepted too (hipHostMalloc'ed memory is mapped into the GPU's address space: a kernel
    reads it over PCIe - the batch feeder's zero-copy gather)."""
    if t is None:
        if allow_none:
            return None, 0, 0
        raise ValueError(f"{name} is required")
    on_dev = t.is_cuda or (allow_pinned and t.is_pinned())
    if t.dtype != F32 or t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1) or not on_dev:
        raise ValueError(f"{name}: expected a 2-D float32 device tensor with unit column stride, got "
                         f"{t.dtype} {tuple(t.shape)} strides {t.stride()}")
    return t.data_ptr(), t.stride(0), t.shape[1]


def dense_filter_workspace(K, N):
    return int(_lib.lib().rm_dense_filter_workspace(int(K), int(N)))


def dense_wgrad_workspace(K, N, M):
    return int(_lib.lib().rm_dense_wgrad_workspace(int(K), int(N), int(M)))


def dense6_workspace(K, N, M):
    """Floats of workspace for the bf16x6 form of dense_fwd (rm_dense6_workspace)."""
    return int(_lib.lib().rm_dense6_workspace(int(K), int(N), int(M)))


def dense_fwd6_supported(a1, a2, epilogue=DENSE_BIAS_ACT, aux2=None, out2=None):
    """Does rm_dense_fwd6 (fp32 operands split into bf16 pieces, csrc/gemm6.hip) take this call?"""
    K1 = a1.shape[1]
    K2 = 0 if a2 is None else a2.shape[1]
    return (epilogue in (DENSE_BIAS_ACT, DENSE_MUL_ACTGRAD, DENSE_ADD) and aux2 is None and out2 is None
            and a1.stride(1) == 1 and a1.stride(0) % 4 == 0 and a1.data_ptr() % 16 == 0 and K1 % 32 + K2 <= 32)


def dense_fwd(a1, a2, W, out, filter_ws, *, transposed=False, bias=None, epilogue=DENSE_BIAS_ACT,
              act="identity", aux1=None, aux2=None, out2=None, ws6=None, dot=None):
    """out[M,N] = epilogue([a1 | a2] @ (W.T if transposed else W)) (rm_dense_fwd).  ws6 (dense6_workspace floats):
    the call runs on the bf16 matrix pipe with split operands (rm_dense_fwd6) when that kernel takes it; returns
    True when it did.  dot = (w [N], w0 [1] or None, out [M]): with rm_dense_fwd6 also out[b] = out_row(b) . w + w0
    (otherwise the caller runs rowdot)."""
    if ws6 is not None and dense_fwd6_supported(a1, a2, epilogue, aux2, out2):
        p1, lda1, K1 = _rows2d(a1, "a1")
        p2, lda2, K2 = _rows2d(a2, "a2", allow_none=True)
        M = a1.shape[0]
        pw, ldw, wc = _rows2d(W, "W")
        K = K1 + K2
        N = W.shape[0] if transposed else wc
        if (wc if transposed else W.shape[0]) != K:
            raise ValueError(f"W {tuple(W.shape)} does not match K={K} (transposed={transposed})")
        pc, ldc, nc = _rows2d(out, "out")
        if out.shape[0] != M or nc != N:
            raise ValueError(f"out {tuple(out.shape)} must be [{M},{N}]")
        px1, ld1, _ = _rows2d(aux1, "aux1", allow_none=True)
        if aux1 is not None and tuple(aux1.shape) != (M, N):
            raise ValueError(f"aux1 {tuple(aux1.shape)} must be [{M},{N}]")
        if ws6.numel() < dense6_workspace(K, N, M):
            raise ValueError("ws6 too small (rm_dense6_workspace)")
        dw, dw0, dout = dot if dot is not None else (None, None, None)
        _lib.call("rm_dense_fwd6", p1, lda1, K1, p2, lda2, K2, pw, ldw, int(bool(transposed)), N,
                  _chk(bias, "bias", F32, (N,), allow_none=True), int(epilogue), ACT_IDS[act], px1, ld1, M, pc, ldc,
                  _chk(dw, "dot w", F32, (N,), allow_none=True), _chk(dw0, "dot w0", F32, (1,), allow_none=True),
                  _chk(dout, "dot out", F32, (M,), allow_none=True), _chk(ws6, "ws6", F32), _stream())
        return True
    p1, lda1, K1 = _rows2d(a1, "a1")
    p2, lda2, K2 = _rows2d(a2, "a2", allow_none=True)
    M = a1.shape[0]
    if a2 is not None and a2.shape[0] != M:
        raise ValueError("a1 and a2 differ in rows")
    pw, ldw, wc = _rows2d(W, "W")
    K = K1 + K2
    N = W.shape[0] if transposed else wc
    if (wc if transposed else W.shape[0]) != K:
        raise ValueError(f"W {tuple(W.shape)} does not match K={K} (transposed={transposed})")
    pc, ldc, nc = _rows2d(out, "out")
    if out.shape[0] != M or nc != N:
        raise ValueError(f"out {tuple(out.shape)} must be [{M},{N}]")
    px1, ld1, _ = _rows2d(aux1, "aux1", allow_none=True)
    px2, ld2, _ = _rows2d(aux2, "aux2", allow_none=True)
    pc2, ldc2, _ = _rows2d(out2, "out2", allow_none=True)
    for t, nm in ((aux1, "aux1"), (aux2, "aux2"), (out2, "out2")):
        if t is not None and tuple(t.shape) != (M, N):
            raise ValueError(f"{nm} {tuple(t.shape)} must be [{M},{N}]")
    if filter_ws.numel() < dense_filter_workspace(K, N):
        raise ValueError("filter_ws too small (rm_dense_filter_workspace)")
    _lib.call("rm_dense_fwd", p1, lda1, K1, p2, lda2, K2, pw, ldw, int(bool(transposed)), N,
              _chk(bias, "bias", F32, (N,), allow_none=True), int(epilogue), ACT_IDS[act], px1, ld1, px2, ld2,
              M, pc, ldc, pc2, ldc2, _chk(filter_ws, "filter_ws", F32), _stream())


def dense_wgrad6_workspace(K, N, M):
    """Floats of workspace for the bf16x6 form of dense_wgrad (rm_dense_wgrad6_workspace)."""
    return int(_lib.lib().rm_dense_wgrad6_workspace(int(K), int(N), int(M)))


def dense_wgrad(a1, a2, G, dW, ws, accumulate=False, db=None, ws6=None, G2=None, dW2=None):
    """dW[K,N] (+)= [a1 | a2].T @ G (rm_dense_wgrad); db [N] = G.sum(0) when given.  ws6 (dense_wgrad6_workspace
    floats): on the bf16 matrix pipe with split operands (rm_dense_wgrad6); there G2 [M,N2] / dW2 [K,N2] add a second
    piece of gradient columns to the same pass (ws6 sized for N + N2)."""
    p1, lda1, K1 = _rows2d(a1, "a1")
    p2, lda2, K2 = _rows2d(a2, "a2", allow_none=True)
    pg, ldg, N = _rows2d(G, "G")
    M = a1.shape[0]
    if G.shape[0] != M or (a2 is not None and a2.shape[0] != M):
        raise ValueError("a1 / a2 / G differ in rows")
    pd, lddw, nd = _rows2d(dW, "dW")
    if dW.shape[0] != K1 + K2 or nd != N:
        raise ValueError(f"dW {tuple(dW.shape)} must be [{K1 + K2},{N}]")
    if G2 is not None and ws6 is None:
        raise ValueError("dense_wgrad: G2 needs the ws6 path")
    if ws6 is not None:
        pg2, ldg2, N2 = _rows2d(G2, "G2", allow_none=True)
        pd2, lddw2, nd2 = _rows2d(dW2, "dW2", allow_none=True)
        if G2 is not None and (G2.shape[0] != M or dW2 is None or dW2.shape[0] != K1 + K2 or nd2 != N2):
            raise ValueError("dense_wgrad: G2 / dW2 shape mismatch")
        _lib.call("rm_dense_wgrad6", p1, lda1, K1, p2, lda2, K2, pg, ldg, N, pg2, ldg2, N2, M, pd, lddw, pd2, lddw2,
                  int(bool(accumulate)), _chk(db, "db", F32, (N,), allow_none=True), _chk(ws6, "ws6", F32),
                  ws6.numel(), _stream())
        return
    _lib.call("rm_dense_wgrad", p1, lda1, K1, p2, lda2, K2, pg, ldg, N, M, pd, lddw, int(bool(accumulate)),
              _chk(db, "db", F32, (N,), allow_none=True), _chk(ws, "ws", F32), ws.numel(), _stream())
