"""GPU parity of the CIN MFMA kernels against the CPU oracle (float64 twin for the
contraction, so the comparison prices only the kernel's own fp32 rounding)."""
import numpy as np
import pytest
import torch

from oracle import th_layers as T

pytestmark = pytest.mark.gpu


def _layer_ref(X0, Xk, W, bias, act):
    B, m, D = X0.shape
    Z = torch.einsum("bid,bjd->bdij", X0.double(), Xk.double()).reshape(B, D, -1)
    M = Z @ W.double() + bias.double()
    return T.act_fn(act)(M).transpose(1, 2).contiguous()  # [B,N,D]


@pytest.mark.parametrize("B,m,H,N,D,act", [
    (5, 2, 2, 16, 4, "identity"),
    (33, 5, 5, 12, 8, "leaky_relu"),
    (40, 26, 26, 128, 16, "leaky_relu"),
    (40, 26, 64, 128, 16, "leaky_relu"),
    (17, 7, 3, 40, 16, "relu"),        # odd H -> padded k', N not a multiple of 32
    (9, 6, 25, 100, 32, "leaky_relu"),  # the reference default units 100 -> H = 50/25
    (6, 4, 8, 64, 64, "relu"),         # D = 64: one example spans both M-tiles
])
def test_cin_layer_fwd(hip_lib, B, m, H, N, D, act):
    from recman_amd import ops

    g = torch.Generator().manual_seed(B * 131 + N)
    X0 = torch.randn(B, m, D, generator=g)
    Hk = H + 3  # Xk is the first H rows of a wider map
    Xkfull = torch.randn(B, Hk, D, generator=g)
    W = torch.randn(m * H, N, generator=g) * 0.2
    bias = torch.randn(N, generator=g) * 0.1
    want = _layer_ref(X0, Xkfull[:, :H], W, bias, act)
    pool_from = N // 2
    want_pool = want[:, pool_from:].sum(-1)
    out = torch.empty(B, N, D, device="cuda")
    pooled = torch.full((B, 7 + N - pool_from), -1.0, device="cuda")
    ws = torch.empty(ops.cin_filter_workspace(m, H, N), device="cuda")
    ops.cin_layer_fwd(X0.cuda(), Xkfull.cuda(), H, W.cuda(), bias.cuda(), act, out, ws,
                      pooled=pooled, pool_col0=7, pool_from=pool_from)
    torch.cuda.synchronize()
    scale = float(want.abs().max())
    err = float((out.cpu().double() - want).abs().max())
    assert err <= 2e-6 * max(1.0, scale) * max(1, (m * H) ** 0.5 / 4), (err, scale)
    perr = float((pooled[:, 7:].cpu().double() - want_pool).abs().max())
    assert perr <= 1e-5 * max(1.0, float(want_pool.abs().max())), perr
    assert bool((pooled[:, :7] == -1).all())  # untouched columns


def test_cin_notebook_kat_on_gpu(hip_lib):
    # recman/notes/xDeepFM.ipynb cell 6 (all-ones filters): exact small integers in fp32
    from recman_amd import ops

    E = torch.tensor([[[1., 2, 3, 4], [5, 6, 7, 8]]]).cuda()
    ws = torch.empty(ops.cin_filter_workspace(2, 8, 16), device="cuda")
    out0 = torch.empty(1, 16, 4, device="cuda")
    pooled = torch.zeros(1, 24, device="cuda")
    ops.cin_layer_fwd(E, E, 2, torch.ones(4, 16).cuda(), torch.zeros(16).cuda(), "identity", out0,
                      ws, pooled=pooled, pool_col0=0, pool_from=8)
    out1 = torch.empty(1, 16, 4, device="cuda")
    ops.cin_layer_fwd(E, out0, 8, torch.ones(16, 16).cuda(), torch.zeros(16).cuda(), "identity",
                      out1, ws, pooled=pooled, pool_col0=8, pool_from=0)
    assert out0[0, 0].tolist() == [36, 64, 100, 144]
    assert out1[0, 0].tolist() == [1728, 4096, 8000, 13824]
    assert pooled[0].tolist() == [344] * 8 + [27648] * 16


@pytest.mark.parametrize("B,m,H,N,D,act,first,last", [
    (5, 2, 2, 16, 4, "identity", True, True),
    (33, 5, 5, 12, 8, "leaky_relu", True, False),
    (21, 26, 26, 128, 16, "leaky_relu", True, False),
    (21, 26, 64, 128, 16, "leaky_relu", False, True),
    (17, 7, 3, 40, 16, "relu", False, False),
    (9, 6, 25, 100, 32, "leaky_relu", False, True),
    (6, 4, 8, 64, 64, "relu", False, False),
    (70, 3, 3, 8, 16, "leaky_relu", True, False),   # several dX blocks and dW chunks
    (6, 4, 4, 64, 64, "relu", True, False),         # first layer (symmetric dX) at D = 64
    (9, 7, 7, 100, 32, "leaky_relu", True, True),   # ... odd m, D = 32, N not a multiple of 32
    (300, 26, 26, 128, 16, "leaky_relu", True, False),  # ... many blocks, ragged last block
    (37, 26, 64, 128, 16, "leaky_relu", False, False),  # configs[2] layer 1 (the split-operand dX kernel's shape)
    (7, 5, 32, 100, 64, "relu", False, True),           # ... one j half, D = 64, N not a multiple of 16
    (13, 3, 64, 128, 32, "identity", False, False),     # ... D = 32
    (11, 6, 50, 100, 16, "leaky_relu", False, False),   # ... H = 50 padded to 64 (the reference's default units 100)
    (40, 26, 26, 128, 16, "leaky_relu", True, False),   # the first layer (H = 26 padded to 32) - "first6" below
    (1, 2, 5, 128, 16, "relu", False, True),            # one example: half a row slab, most of every tile empty
    (3, 1, 1, 70, 64, "leaky_relu", False, False),      # m = H = 1 (not Xk = X0), N = 70 (> 64: the split dX / dW take it)
    (129, 4, 33, 128, 32, "identity", False, False),    # H = 33 padded to 64; a last block with one example
])
@pytest.mark.parametrize("split", [False, True, "first6"], ids=["f32", "split", "split-first-layer-too"])
def test_cin_layer_bwd(hip_lib, B, m, H, N, D, act, first, last, split):
    """autograd (float64) of the oracle's layer formula vs rm_cin_layer_bwd; split: the dX pass on the bf16 matrix
    pipe with split fp32 operands where csrc/cin6.hip covers the layer (elsewhere the flag changes nothing)."""
    from recman_amd import ops

    g_ = torch.Generator().manual_seed(B * 7 + N)
    X0 = torch.randn(B, m, D, generator=g_, dtype=torch.float64, requires_grad=True)
    if first:
        Xk = X0
    else:
        Xk = torch.randn(B, H, D, generator=g_, dtype=torch.float64, requires_grad=True)
    W = (torch.randn(m * H, N, generator=g_, dtype=torch.float64) * 0.2).requires_grad_(True)
    bias = (torch.randn(N, generator=g_, dtype=torch.float64) * 0.1).requires_grad_(True)
    pool_from = 0 if last else N // 2
    cw = torch.randn(N - pool_from, generator=g_, dtype=torch.float64)
    gvec = torch.randn(B, generator=g_, dtype=torch.float64)
    dh = torch.randn(B, pool_from, D, generator=g_, dtype=torch.float64) if pool_from else None
    Z = torch.einsum("bid,bjd->bdij", X0, Xk).reshape(B, D, -1)
    out = T.act_fn(act)(Z @ W + bias).transpose(1, 2)  # [B,N,D]
    obj = (out[:, pool_from:].sum(-1) * cw * gvec[:, None]).sum()
    if pool_from:
        obj = obj + (out[:, :pool_from] * dh).sum()
    obj.backward()

    f = lambda t: t.detach().float().cuda().contiguous()
    out_d = f(out)
    dX0 = torch.full((B, m, D), 0.5, device="cuda")  # accumulate onto a known value
    dXk = None if first else torch.empty(B, H, D, device="cuda")
    dW = torch.empty(m * H, N, device="cuda")
    dbias = torch.empty(N, device="cuda")
    ws = torch.empty(ops.cin_bwd_workspace(B, m, H, N, D), device="cuda")
    ops.cin_layer_bwd(f(X0), f(Xk), H, f(W), act, out_d, f(gvec), dX0, dW, dbias, ws,
                      xk_is_x0=first, d_hidden=f(dh) if pool_from else None, cin_w_direct=f(cw),
                      pool_from=pool_from, accumulate_dx0=True, dXk=dXk, split=bool(split), first6=split == "first6")
    torch.cuda.synchronize()

    def close(got, want, what):
        want = want.double()
        scale = max(1.0, float(want.abs().max()))
        err = float((got.cpu().double() - want).abs().max())
        assert err <= 2e-5 * scale, f"{what}: {err:.3e} (scale {scale:.3e})"

    close(dX0 - 0.5, X0.grad, "dX0")
    if not first:
        close(dXk, Xk.grad, "dXk")
    close(dW, W.grad, "dW")
    close(dbias, bias.grad, "dbias")


@pytest.mark.parametrize("B,m,H,N,D,act", [
    (40, 26, 64, 128, 16, "leaky_relu"),   # configs[2] layer 1
    (9, 26, 64, 128, 64, "leaky_relu"),    # configs[4] layer 1: an example spans four row tiles over two waves
    (21, 7, 32, 48, 32, "relu"),           # one j half, N < 128 (zero filter tiles), two examples per wave
    (3, 3, 64, 100, 16, "identity"),       # N not a multiple of 16, a block with 3 of its 8 examples
    (1, 1, 32, 16, 64, "leaky_relu"),
    (11, 6, 50, 100, 16, "leaky_relu"),    # H = 50 padded to 64
    (9, 5, 3, 40, 32, "relu"),             # H = 3 padded to 32
])
def test_cin_layer_fwd6_split_operands(hip_lib, B, m, H, N, D, act):
    """rm_cin_layer_fwd6 (csrc/cin6.hip: Z = fl(x0 * xk) split into three bf16 pieces, six piece products on the
    bf16 matrix pipe) against the float64 layer to the f32 kernel's tolerance, never worse than 1.5x the f32-MFMA
    kernel's own error; pooled columns, untouched columns, determinism; and what it declines."""
    from recman_amd import ops

    g = torch.Generator().manual_seed(B * 17 + N)
    X0 = torch.randn(B, m, D, generator=g)
    Xkfull = torch.randn(B, H + 3, D, generator=g)
    W = torch.randn(m * H, N, generator=g) * 0.2
    bias = torch.randn(N, generator=g) * 0.1
    want = _layer_ref(X0, Xkfull[:, :H], W, bias, act)
    pool_from = N // 2
    want_pool = want[:, pool_from:].sum(-1)
    need = ops.cin_filter_workspace6(m, H, N, D)
    assert need > 0
    ws = torch.empty(ops.cin_filter_workspace(m, H, N), device="cuda")
    ws6 = torch.empty(need, device="cuda")
    outs = []
    for use6 in (True, False, True):
        out = torch.full((B, N, D), float("nan"), device="cuda")
        pooled = torch.full((B, 7 + N - pool_from), -1.0, device="cuda")
        took = ops.cin_layer_fwd(X0.cuda(), Xkfull.cuda(), H, W.cuda(), bias.cuda(), act, out, ws, pooled=pooled,
                                 pool_col0=7, pool_from=pool_from, ws6=ws6 if use6 else None)
        assert bool(took) == use6
        outs.append((out, pooled))
    scale = float(want.abs().max())
    e6 = float((outs[0][0].cpu().double() - want).abs().max())
    e32 = float((outs[1][0].cpu().double() - want).abs().max())
    assert e6 <= 2e-6 * max(1.0, scale) * max(1, (m * H) ** 0.5 / 4), (e6, scale)
    assert e6 <= 1.5 * e32 + 1e-7, (e6, e32)
    perr = float((outs[0][1][:, 7:].cpu().double() - want_pool).abs().max())
    assert perr <= 1e-5 * max(1.0, float(want_pool.abs().max())), perr
    assert bool((outs[0][1][:, :7] == -1).all())
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    # not covered: D = 8, H > 64, N > 128
    assert ops.cin_filter_workspace6(26, 64, 128, 8) == 0
    assert ops.cin_filter_workspace6(26, 96, 128, 16) == 0 and ops.cin_filter_workspace6(26, 64, 200, 16) == 0
