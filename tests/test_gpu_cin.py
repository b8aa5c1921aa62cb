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
