"""GPU: the fused CrossNet kernels (rm_cross_fwd / rm_cross_bwd / rm_cross_param_grads, csrc/cross.hip)
through the C ABI against the oracle's restatement of arXiv 1708.05123 eq. (3) (oracle/th_layers.cross_net;
the class is absent from the reference, DCN.py:7,134-137) and torch autograd of it, in float64.

The kernels work in closed form (x_l = c_l x0 + sum_{j<l} b_j: L+1 dot products + a scalar recurrence, the
backward from the saved dot products alone), so these cases pin that algebra: every lane mapping the
kernels template on (FD <= 256 / <= 512, no / <= 64 / > 64 dense columns), row counts around the
rows-per-wave unroll, L = 1..8, and the gradient hand-off to the parameter-gradient GEMM."""
import pytest
import torch

from oracle import th_layers as T

pytestmark = pytest.mark.gpu


def _case(B, FD, Dn, L, seed=0, scale=0.3):
    g = torch.Generator().manual_seed(seed)
    d = FD + Dn
    xe = torch.randn(B, FD, generator=g)
    xd = torch.randn(B, Dn, generator=g) if Dn else None
    w = torch.randn(L, d, generator=g) * (scale / d ** 0.5)
    b = torch.randn(L, d, generator=g) * 0.1
    wo = torch.randn(d, generator=g) * (1.0 / d ** 0.5)
    gl = torch.randn(B, generator=g)
    dxin = torch.randn(B, FD, generator=g)
    return xe, xd, w, b, wo, gl, dxin


def _oracle(xe, xd, w, b, wo, gl):
    """float64 autograd of the layer-by-layer recurrence: logit, d/dxe, d/dw, d/db, d/dw_out of sum(g*logit)."""
    x0 = torch.cat([xe] + ([xd] if xd is not None else []), 1).double().requires_grad_(True)
    p = {"cross_w": w.double().requires_grad_(True), "cross_b": b.double().requires_grad_(True),
         "cross_w_out": wo.double().view(-1, 1).requires_grad_(True)}
    logit = T.cross_net(p, x0).reshape(-1)
    (logit * gl.double()).sum().backward()
    return logit.detach(), x0.grad[:, : xe.shape[1]], p["cross_w"].grad, p["cross_b"].grad, p["cross_w_out"].grad.view(-1)


def _close(got, want, rtol, what):
    err = float((got.double().cpu() - want).abs().max())
    scale = max(1e-6, float(want.abs().max()))
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("B,FD,Dn,L", [
    (37, 40, 3, 3),        # FD <= 256: one slice per lane; a handful of dense columns
    (64, 256, 0, 2),       # no dense part, FD exactly one full slice
    (130, 416, 13, 6),     # the Criteo shape (BASELINE configs[3]): 104 slices, 13 dense columns
    (9, 512, 64, 8),       # the largest embedding block, a full lane row of dense columns, L = 8
    (33, 24, 100, 1),      # more than 64 dense columns (the dynamic dense loop), L = 1
    (1, 8, 1, 4), (2, 8, 1, 4), (3, 8, 1, 4),  # row counts around the rows-per-wave unroll
    (4100, 416, 13, 6),    # more rows than one pass of the grid's waves
])
def test_cross_fwd_bwd_match_autograd(hip_lib, B, FD, Dn, L):
    from recman_amd import ops

    xe, xd, w, b, wo, gl, dxin = _case(B, FD, Dn, L)
    logit_o, dxe_o, dw_o, db_o, dwo_o = _oracle(xe, xd, w, b, wo, gl)
    dev = "cuda"
    c = lambda t: None if t is None else t.to(dev).contiguous()  # noqa: E731
    xe_, xd_, w_, b_, wo_, gl_, dxin_ = map(c, (xe, xd, w, b, wo, gl, dxin))
    logit = torch.empty(B, device=dev)
    p = torch.full((B, ops.cross_p_ld(L)), float("nan"), device=dev)
    ops.cross_fwd(xe_, xd_, w_, b_, wo_, logit, p)
    _close(logit, logit_o, 2e-6, "logit")
    x0 = torch.cat([xe] + ([xd] if xd is not None else []), 1).double()
    want_p = torch.cat([x0 @ w.double().t(), (x0 @ wo.double()).view(-1, 1)], 1)
    _close(p[:, : L + 1], want_p, 2e-6, "saved dot products")
    # inference form: no p row
    logit2 = torch.empty(B, device=dev)
    ops.cross_fwd(xe_, xd_, w_, b_, wo_, logit2, None)
    assert torch.equal(logit2, logit)

    for dx in (dxin_, None):
        d_xe = torch.full((B, FD), float("nan"), device=dev)
        coef = torch.full((B, 2 * L + 2), float("nan"), device=dev)
        ops.cross_bwd(w_, b_, wo_, gl_, p, d_xe, coef, dx_in_e=dx)
        want = dxe_o + (dxin.double() if dx is not None else 0)
        _close(d_xe, want, 5e-6, "d_xe")
        assert torch.isfinite(coef).all()
        _close(coef[:, 2 * L + 1], gl.double(), 0, "coef g column")
    # parameter gradients: P = x0^T coef[:, :L+1], column sums of the rest (what the engine does with
    # rm_dense_wgrad / rm_linear_dense_bwd; plain torch here - the hand-off is what is under test)
    P = (x0.t() @ coef[:, : L + 1].double().cpu()).float().to(dev).contiguous()
    colsum = coef[:, L + 1:].double().sum(0).float().contiguous()
    d = FD + Dn
    dw, db, dwo = (torch.empty(L, d, device=dev), torch.empty(L, d, device=dev), torch.empty(d, device=dev))
    ops.cross_param_grads(P, colsum, w_, b_, wo_, dw, db, dwo)
    _close(dw, dw_o, 2e-5, "d cross_w")
    _close(db, db_o, 2e-5, "d cross_b")
    _close(dwo, dwo_o, 2e-5, "d cross_w_out")


def test_cross_kernels_are_deterministic_and_reject_bad_shapes(hip_lib):
    from recman_amd import _lib, ops

    B, FD, Dn, L = 777, 416, 13, 6
    xe, xd, w, b, wo, gl, dxin = (t.cuda() if t is not None else None for t in _case(B, FD, Dn, L, seed=3))
    outs = []
    for _ in range(2):
        logit, p = torch.empty(B, device="cuda"), torch.zeros(B, ops.cross_p_ld(L), device="cuda")
        d_xe, coef = torch.empty(B, FD, device="cuda"), torch.empty(B, 2 * L + 2, device="cuda")
        ops.cross_fwd(xe, xd, w, b, wo, logit, p)
        ops.cross_bwd(w, b, wo, gl, p, d_xe, coef, dx_in_e=dxin)
        outs.append((logit, p, d_xe, coef))
    for a, c in zip(*outs):
        assert torch.equal(a, c)
    with pytest.raises(_lib.RecmanHipError):   # L > 8
        ops.cross_fwd(xe, xd, torch.zeros(9, FD + Dn, device="cuda"), torch.zeros(9, FD + Dn, device="cuda"), wo,
                      torch.empty(B, device="cuda"), None)
    with pytest.raises(_lib.RecmanHipError):   # FD not a multiple of 4
        ops.cross_fwd(torch.zeros(4, 6, device="cuda"), None, torch.zeros(1, 6, device="cuda"),
                      torch.zeros(1, 6, device="cuda"), torch.zeros(6, device="cuda"), torch.empty(4, device="cuda"))
    with pytest.raises(_lib.RecmanHipError):   # p row shorter than L + 1
        ops.cross_fwd(xe, xd, w, b, wo, torch.empty(B, device="cuda"), torch.zeros(B, 4, device="cuda"))
