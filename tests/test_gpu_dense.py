"""GPU: the wide dense-layer kernels (rm_dense_fwd / rm_dense_wgrad, csrc/gemm.hip) against a
float64 torch reference of the same op.  Tolerance: f32 accumulation over K (or over the
batch for wgrad) - 2e-5 relative to the output scale."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_act(x, act):
    if act == "relu":
        return torch.relu(x)
    if act == "leaky_relu":
        return torch.nn.functional.leaky_relu(x, 0.2)
    return x


def _close(got, want, tol=2e-5, what=""):
    want = want.to(torch.float64)
    scale = max(1.0, float(want.abs().max()))
    err = float((got.detach().cpu().to(torch.float64) - want).abs().max())
    assert err <= tol * scale, f"{what}: max err {err:.3e} (scale {scale:.3e})"


SHAPES = [  # M, K1, K2, N
    (37, 40, 3, 50),        # everything ragged, one column tile, NTMAX=2
    (300, 416, 13, 400),    # DCN layer 0: x = [E | dense], 13 column blocks (7 + 6)
    (129, 400, 0, 400),     # DCN layer 1, M tail of one row
    (64, 429, 0, 429),      # matrix cross: K tail inside A1 (lda 432), 14 column blocks
    (200, 64, 0, 900),      # three column tiles (14 + 14 + 1 blocks)
    (70, 24, 5, 200),       # NTMAX=4
]


def _dev_padded(a1):
    """Device copy of [M,K1] whose row stride is padded to a multiple of 4 floats."""
    M, K1 = a1.shape
    buf = torch.zeros(M, (K1 + 3) // 4 * 4, device="cuda")
    buf[:, :K1] = a1.cuda()
    return buf[:, :K1]


def _inputs(M, K1, K2, N, seed=0):
    g = torch.Generator().manual_seed(seed)
    lda1 = (K1 + 3) // 4 * 4
    a1 = torch.randn(M, lda1, generator=g)[:, :K1]
    a2 = torch.randn(M, K2, generator=g) if K2 else None
    W = torch.randn(K1 + K2, N, generator=g) / (K1 + K2) ** 0.5
    bias = torch.randn(N, generator=g)
    return a1, a2, W, bias


@pytest.mark.parametrize("M,K1,K2,N", SHAPES)
@pytest.mark.parametrize("act", ["relu", "leaky_relu", "identity"])
def test_dense_fwd_bias_act(hip_lib, M, K1, K2, N, act):
    from recman_amd import ops

    a1, a2, W, bias = _inputs(M, K1, K2, N)
    x = torch.cat([a1] + ([a2] if a2 is not None else []), dim=1).double()
    want = _ref_act(x @ W.double() + bias.double(), act)
    a1d = _dev_padded(a1)
    assert a1d.stride(0) % 4 == 0
    out = torch.full((M, N), float("nan"), device="cuda")
    ws = torch.empty(ops.dense_filter_workspace(K1 + K2, N), device="cuda")
    ops.dense_fwd(a1d, a2.cuda() if a2 is not None else None, W.cuda(), out, ws, bias=bias.cuda(), act=act)
    _close(out, want, what="bias_act")
    # op(W) = W^T given as [N, K]
    out.fill_(float("nan"))
    ops.dense_fwd(a1d, a2.cuda() if a2 is not None else None, W.t().contiguous().cuda(), out, ws,
                  transposed=True, bias=bias.cuda(), act=act)
    _close(out, want, what="bias_act, transposed W")


@pytest.mark.parametrize("M,K1,K2,N", SHAPES[:4])
def test_dense_fwd_other_epilogues(hip_lib, M, K1, K2, N):
    from recman_amd import ops

    a1, a2, W, bias = _inputs(M, K1, K2, N, seed=1)
    g = torch.Generator().manual_seed(5)
    aux1, aux2 = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    x = torch.cat([a1] + ([a2] if a2 is not None else []), dim=1).double()
    z = x @ W.double()
    a1d, a2d = _dev_padded(a1), (a2.cuda() if a2 is not None else None)
    ws = torch.empty(ops.dense_filter_workspace(K1 + K2, N), device="cuda")
    out = torch.empty(M, N, device="cuda")
    # dX of a hidden layer: (dh @ W^T) * act'(h) from the post-activation h
    ops.dense_fwd(a1d, a2d, W.cuda(), out, ws, epilogue=ops.DENSE_MUL_ACTGRAD, act="leaky_relu",
                  aux1=aux1.cuda())
    _close(out, z * torch.where(aux1 > 0, 1.0, 0.2).double(), what="mul_actgrad")
    ops.dense_fwd(a1d, a2d, W.cuda(), out, ws, epilogue=ops.DENSE_ADD, aux1=aux1.cuda())
    _close(out, z + aux1.double(), what="add")
    ops.dense_fwd(a1d, a2d, W.cuda(), out, ws, epilogue=ops.DENSE_ADD)
    _close(out, z, what="plain")
    # padded output / aux rows (ld > N)
    outp = torch.zeros(M, N + 7, device="cuda")
    u = torch.empty(M, N, device="cuda")
    auxp = torch.zeros(M, N + 5)
    auxp[:, :N] = aux1
    ops.dense_fwd(a1d, a2d, W.cuda(), outp[:, :N], ws, bias=bias.cuda(), epilogue=ops.DENSE_CROSS,
                  aux1=auxp.cuda()[:, :N], aux2=aux2.cuda(), out2=u)
    uu = z + bias.double()
    _close(u, uu, what="cross u")
    _close(outp[:, :N], aux1.double() * uu + aux2.double(), what="cross")
    assert float(outp[:, N:].abs().max()) == 0.0


@pytest.mark.parametrize("M,K1,K2,N", SHAPES + [(5000, 416, 13, 400), (3, 8, 0, 8),
                                                # skinny G (N <= 8): the weighted-column-sum path
                                                (5000, 416, 13, 7), (129, 32, 0, 1), (4100, 64, 3, 8),
                                                (300, 30, 2, 5),    # K1 % 4 != 0: tiled path
                                                # ragged last K tile split over the block's SIMDs
                                                (5000, 400, 0, 400), (4100, 272, 0, 96), (9000, 429, 0, 200),
                                                (3000, 160, 13, 64)])
def test_dense_wgrad(hip_lib, M, K1, K2, N):
    from recman_amd import ops

    a1, a2, _, _ = _inputs(M, K1, K2, N, seed=2)
    g = torch.Generator().manual_seed(7)
    G = torch.randn(M, N, generator=g)
    x = torch.cat([a1] + ([a2] if a2 is not None else []), dim=1).double()
    want = x.t() @ G.double()
    K = K1 + K2
    ws = torch.empty(ops.dense_wgrad_workspace(K, N, M), device="cuda")
    dW = torch.full((K, N), float("nan"), device="cuda")
    a1d, a2d = _dev_padded(a1), (a2.cuda() if a2 is not None else None)
    ops.dense_wgrad(a1d, a2d, G.cuda(), dW, ws)
    _close(dW, want, tol=3e-5, what="wgrad")
    first = dW.clone()
    ops.dense_wgrad(a1d, a2d, G.cuda(), dW, ws)
    assert torch.equal(dW, first)  # deterministic
    ops.dense_wgrad(a1d, a2d, G.cuda(), dW, ws, accumulate=True)
    _close(dW, 2 * want, tol=3e-5, what="wgrad accumulate")
    # db: the column sums of G (the layer's bias gradient) from the same launch
    db = torch.full((N,), float("nan"), device="cuda")
    ops.dense_wgrad(a1d, a2d, G.cuda(), dW, ws, db=db)
    _close(dW, want, tol=3e-5, what="wgrad with db")
    _close(db, G.double().sum(0), tol=3e-5, what="db")


def test_dense_fwd_unaligned_rows_and_k1(hip_lib):
    """Rows that are not 16-byte aligned (hidden width 30) and the K = 1 outer product of the
    output projection's backward take the per-element loader."""
    from recman_amd import ops

    g = torch.Generator().manual_seed(3)
    a = torch.randn(77, 30, generator=g)
    W = torch.randn(30, 45, generator=g)
    ws = torch.empty(ops.dense_filter_workspace(30, 45), device="cuda")
    out = torch.empty(77, 45, device="cuda")
    ops.dense_fwd(a.cuda(), None, W.cuda(), out, ws, act="relu")
    _close(out, torch.relu(a.double() @ W.double()), what="unaligned rows")
    gv, w = torch.randn(77, 1, generator=g), torch.randn(45, 1, generator=g)
    h = torch.randn(77, 45, generator=g)
    ops.dense_fwd(gv.cuda(), None, w.cuda(), out, ws, transposed=True, epilogue=ops.DENSE_MUL_ACTGRAD,
                  act="relu", aux1=h.cuda())
    _close(out, (gv.double() @ w.double().t()) * (h > 0).double(), what="K=1 outer product")


def test_dense_argument_checks(hip_lib):
    from recman_amd import ops
    from recman_amd._lib import RecmanHipError

    a = torch.randn(8, 8, device="cuda")
    W = torch.randn(8, 8, device="cuda")
    out = torch.empty(8, 8, device="cuda")
    ws = torch.empty(ops.dense_filter_workspace(8, 8), device="cuda")
    with pytest.raises(ValueError):
        ops.dense_fwd(a, None, torch.randn(9, 8, device="cuda"), out, ws)
    with pytest.raises(ValueError):
        ops.dense_fwd(a, None, W, out, ws[:10])
    with pytest.raises(RecmanHipError):
        ops.dense_fwd(a, None, W, out, ws, epilogue=ops.DENSE_CROSS)  # needs aux1/aux2
    with pytest.raises(RecmanHipError):
        ops.dense_fwd(a, None, W, out, ws, epilogue=ops.DENSE_MUL_ACTGRAD)  # needs aux1


@pytest.mark.parametrize("B,N,act", [(1, 64, "relu"), (257, 400, "relu"), (5000, 128, "leaky_relu"),
                                      (131, 68, "identity")])
def test_outer_actgrad_sums(hip_lib, B, N, act):
    """rm_outer_actgrad_sums: da = g w^T o act'(a) plus the three column reductions, against torch
    in float64."""
    from recman_amd import ops

    gen = torch.Generator().manual_seed(B * 7 + N)
    g = torch.randn(B, generator=gen)
    w = torch.randn(N, generator=gen)
    a = torch.randn(B, N, generator=gen)
    if act == "relu":
        a = a.clamp_min(0)
    slope = {"relu": 0.0, "leaky_relu": 0.2, "identity": 1.0}[act]
    fac = torch.where(a > 0, 1.0, slope) if act != "identity" else torch.ones_like(a)
    want_da = (g.double()[:, None] * w.double()[None, :]) * fac.double()
    gd, wd, ad = g.cuda(), w.cuda(), a.cuda()
    da = torch.empty(B, N, device="cuda")
    d_w, d_w0, db = torch.empty(N, device="cuda"), torch.empty(1, device="cuda"), torch.empty(N, device="cuda")
    ws = torch.empty(ops.outer_actgrad_sums_workspace(B, N), device="cuda")
    ops.outer_actgrad_sums(gd, wd, ad, act, da, d_w, d_w0, db, ws)
    ref = torch.empty(B, N, device="cuda")
    ops.outer_actgrad(gd, wd, ad, act, ref)
    assert torch.equal(da, ref)
    _close(da, want_da.float())
    _close(d_w, (g.double()[:, None] * a.double()).sum(0).float())
    _close(d_w0, g.double().sum().reshape(1).float())
    _close(db, want_da.sum(0).float())


SHAPES6 = [  # M, K1, K2, N: what rm_dense_fwd6 takes (K1 % 32 + K2 <= 32)
    (300, 416, 13, 400),    # DCN layer 0: 13 whole slabs of E + the dense inputs as a padded copy
    (129, 400, 0, 400),     # DCN layer 1: ragged K, the last slab re-reads [368, 400) in place; M tail
    (777, 400, 0, 416),     # DCN's dX0 (transposed weights below), two full column groups, 4 row tiles
    (37, 40, 3, 50),        # everything ragged, one column group
    (64, 64, 0, 900),       # five column groups
    (5, 8, 0, 8),           # less than a slab: all of it through the copy
]


@pytest.mark.parametrize("M,K1,K2,N", SHAPES6)
def test_dense_fwd6_split_operands_match_float64_at_least_as_well_as_the_f32_kernel(hip_lib, M, K1, K2, N):
    """rm_dense_fwd6 (fp32 operands split into three bf16 pieces, six piece products on the bf16 matrix pipe, fp32
    accumulate): every epilogue it covers, both weight layouts and the fused output projection against float64 -
    to the f32 kernel's tolerance, and never worse than 1.5x the f32-MFMA kernel's own error on the same input."""
    from recman_amd import ops

    a1, a2, W, bias = _inputs(M, K1, K2, N, seed=3)
    g = torch.Generator().manual_seed(6)
    aux1 = torch.randn(M, N, generator=g)
    wdot, w0 = torch.randn(N, generator=g), torch.randn(1, generator=g)
    x = torch.cat([a1] + ([a2] if a2 is not None else []), dim=1).double()
    z = x @ W.double()
    a1d, a2d = _dev_padded(a1), (a2.cuda() if a2 is not None else None)
    assert ops.dense_fwd6_supported(a1d, a2d)
    ws = torch.empty(ops.dense_filter_workspace(K1 + K2, N), device="cuda")
    ws6 = torch.empty(ops.dense6_workspace(K1 + K2, N, M), device="cuda")
    out6 = torch.full((M, N), float("nan"), device="cuda")
    out32 = torch.empty(M, N, device="cuda")
    dot = torch.full((M,), float("nan"), device="cuda")
    want = torch.relu(z + bias.double())
    took = ops.dense_fwd(a1d, a2d, W.cuda(), out6, ws, bias=bias.cuda(), act="relu", ws6=ws6,
                         dot=(wdot.cuda(), w0.cuda(), dot))
    assert took is True
    ops.dense_fwd(a1d, a2d, W.cuda(), out32, ws, bias=bias.cuda(), act="relu")
    _close(out6, want, what="bias + relu")
    _close(dot, want @ wdot.double() + w0.double(), what="fused output projection")
    e6 = float((out6.cpu().double() - want).abs().max())
    e32 = float((out32.cpu().double() - want).abs().max())
    assert e6 <= 1.5 * e32 + 1e-7, (e6, e32)
    # transposed weights, padded output rows
    outp = torch.zeros(M, N + 4, device="cuda")
    assert ops.dense_fwd(a1d, a2d, W.t().contiguous().cuda(), outp[:, :N], ws, transposed=True, bias=bias.cuda(),
                         act="leaky_relu", ws6=ws6)
    _close(outp[:, :N], torch.nn.functional.leaky_relu(z + bias.double(), 0.2), what="transposed W")
    assert float(outp[:, N:].abs().max()) == 0.0
    # the backward's epilogues
    assert ops.dense_fwd(a1d, a2d, W.cuda(), out6, ws, epilogue=ops.DENSE_MUL_ACTGRAD, act="leaky_relu",
                         aux1=aux1.cuda(), ws6=ws6)
    _close(out6, z * torch.where(aux1 > 0, 1.0, 0.2).double(), what="mul_actgrad")
    assert ops.dense_fwd(a1d, a2d, W.cuda(), out6, ws, epilogue=ops.DENSE_ADD, aux1=aux1.cuda(), ws6=ws6)
    _close(out6, z + aux1.double(), what="add")
    assert ops.dense_fwd(a1d, a2d, W.cuda(), out6, ws, epilogue=ops.DENSE_ADD, ws6=ws6)
    _close(out6, z, what="plain")
    # deterministic
    again = torch.empty_like(out6)
    ops.dense_fwd(a1d, a2d, W.cuda(), again, ws, epilogue=ops.DENSE_ADD, ws6=ws6)
    assert torch.equal(again, out6)


def test_dense_fwd6_declines_what_it_does_not_cover_and_keeps_large_and_tiny_values(hip_lib):
    from recman_amd import ops

    a = torch.randn(16, 80, device="cuda")
    assert not ops.dense_fwd6_supported(a[:, :48], torch.randn(16, 20, device="cuda"))   # ragged end of 36 columns
    assert not ops.dense_fwd6_supported(a[:, 1:65], None)                                 # unaligned rows
    assert not ops.dense_fwd6_supported(a[:, :64], None, ops.DENSE_CROSS)
    # values across the exponent range: the three pieces follow the exponent (bf16 has fp32's range)
    M, K, N = 64, 64, 32
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g) * torch.tensor([1e-18, 1e-6, 1.0, 1e6]).repeat(K // 4)
    W = torch.randn(K, N, generator=g) * torch.tensor([1e12, 1.0, 1e-3, 1e-12]).repeat_interleave(K // 4)[:, None]
    out = torch.empty(M, N, device="cuda")
    ws = torch.empty(ops.dense_filter_workspace(K, N), device="cuda")
    ws6 = torch.empty(ops.dense6_workspace(K, N, M), device="cuda")
    assert ops.dense_fwd(x.cuda(), None, W.cuda(), out, ws, epilogue=ops.DENSE_ADD, ws6=ws6)
    want = x.double() @ W.double()
    # per element against the sum of |products| (the honest scale of a dot product's rounding error)
    scale = (x.double().abs() @ W.double().abs())
    assert float(((out.cpu().double() - want).abs() / scale).max()) < 2e-6


@pytest.mark.parametrize("M,K1,K2,N", SHAPES + [(5000, 416, 13, 400), (3, 8, 0, 8), (4096, 400, 0, 400),
                                               (777, 230, 0, 210)])
def test_dense_wgrad6_split_operands(hip_lib, M, K1, K2, N):
    """rm_dense_wgrad6 (both activations split into bf16 pieces, transposed LDS planes) against float64: dW, the
    accumulate form, the bias gradient; ragged batch (M % 32 != 0), K and N beyond one block tile (224 x 208);
    two runs are bit-identical (partial tiles added in split order)."""
    from recman_amd import ops

    a1, a2, _, _ = _inputs(M, K1, K2, N, seed=2)
    g = torch.Generator().manual_seed(9)
    G = torch.randn(M, N, generator=g)
    x = torch.cat([a1] + ([a2] if a2 is not None else []), dim=1).double()
    want = x.t() @ G.double()
    a1d, a2d = _dev_padded(a1), (a2.cuda() if a2 is not None else None)
    K = K1 + K2
    ws = torch.empty(ops.dense_wgrad_workspace(K, N, M), device="cuda")
    ws6 = torch.empty(ops.dense_wgrad6_workspace(K, N, M), device="cuda")
    dW = torch.full((K, N), float("nan"), device="cuda")
    db = torch.full((N,), float("nan"), device="cuda")
    ops.dense_wgrad(a1d, a2d, G.cuda(), dW, ws, db=db, ws6=ws6)
    tol = 2e-5 * max(1.0, (M / 1000) ** 0.5)   # (fp32 sums over the batch: as the f32 kernel's test)
    _close(dW, want, tol=tol, what="dW")
    _close(db, G.double().sum(0), tol=tol, what="db")
    again = torch.empty_like(dW)
    ops.dense_wgrad(a1d, a2d, G.cuda(), again, ws, ws6=ws6)
    assert torch.equal(again, dW)
    base = torch.randn(K, N + 3, generator=g).cuda()
    acc = base.clone()
    ops.dense_wgrad(a1d, a2d, G.cuda(), acc[:, :N], ws, accumulate=True, ws6=ws6)
    _close(acc[:, :N], want + base[:, :N].cpu().double(), tol=tol, what="accumulate")
    assert torch.equal(acc[:, N:], base[:, N:])
    # a second piece of gradient columns against the same activations in the same pass (DCN's cross coefficients)
    N2 = 7
    G2full = torch.randn(M, 2 * N2 + 2, generator=g)
    ws62 = torch.empty(ops.dense_wgrad6_workspace(K, N + N2, M), device="cuda")
    dW2 = torch.full((K, N2), float("nan"), device="cuda")
    dW.fill_(float("nan"))
    db.fill_(float("nan"))
    ops.dense_wgrad(a1d, a2d, G.cuda(), dW, ws, db=db, ws6=ws62, G2=G2full.cuda()[:, :N2], dW2=dW2)
    _close(dW, want, tol=tol, what="dW beside a second piece")
    _close(dW2, x.t() @ G2full[:, :N2].double(), tol=tol, what="dW2")
    _close(db, G.double().sum(0), tol=tol, what="db beside a second piece")
