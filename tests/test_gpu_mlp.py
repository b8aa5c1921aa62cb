"""GPU parity of the fused skinny-MLP kernels (csrc/mlp.hip) against torch autograd (fp64)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(xe, xd, Ws, bs, w_out, w0, act, g, fm_sum=None):
    from oracle import th_layers as T

    x = torch.cat([xe, xd], 1) if xd is not None else xe
    a = T.act_fn(act)
    hs, y = [], x
    for W, b in zip(Ws, bs):
        y = a(y @ W + b)
        hs.append(y)
    logit = (y @ w_out.reshape(-1, 1)).reshape(-1) + w0
    obj = (logit * g).sum()
    if fm_sum is not None:  # + FM second order through E = xe
        B, FD = xe.shape
        D = fm_sum.shape[1]
        E = xe.reshape(B, FD // D, D)
        s = E.sum(1)
        y2 = 0.5 * (s.square() - E.square().sum(1)).sum(1)
        obj = obj + (y2 * g).sum()
    return logit, hs, obj


@pytest.mark.parametrize("B,FD,Dn,hidden,act,fm", [
    (37, 40, 2, (32,), "relu", False),
    (64, 416, 13, (32, 32), "relu", True),
    (100, 416, 13, (32, 32), "leaky_relu", False),
    (33, 64, 0, (16, 8), "relu", True),
    (70, 128, 5, (24, 32, 7), "leaky_relu", False),
    (5, 8, 1, (3, 2), "identity", False),
    (300, 432, 16, (32, 32), "relu", True),
])
def test_fused_mlp_fwd_bwd(hip_lib, B, FD, Dn, hidden, act, fm):
    from recman_amd import ops

    gen = torch.Generator().manual_seed(B + FD)
    f64 = dict(dtype=torch.float64, generator=gen)
    xe = torch.randn(B, FD, **f64).requires_grad_(True)
    xd = torch.randn(B, Dn, **f64) if Dn else None
    dims = [FD + Dn] + list(hidden)
    Ws = [(torch.randn(dims[i], dims[i + 1], **f64) * 0.2).requires_grad_(True) for i in range(len(hidden))]
    bs = [(torch.randn(dims[i + 1], **f64) * 0.1).requires_grad_(True) for i in range(len(hidden))]
    w_out = (torch.randn(hidden[-1], **f64) * 0.3).requires_grad_(True)
    w0 = torch.randn(1, **f64)
    g = torch.randn(B, **f64)
    D = 8
    S = xe.detach().reshape(B, FD // D, D).sum(1) if fm else None
    logit, hs, obj = _ref(xe, xd, Ws, bs, w_out, w0, act, g, S if fm else None)
    if fm:
        # the FM term's gradient w.r.t. E must treat S as a function of E: recompute with graph
        logit, hs, obj = _ref(xe, xd, Ws, bs, w_out, w0, act, g, torch.zeros(B, D, dtype=torch.float64))
        E = xe.reshape(B, FD // D, D)
        s = E.sum(1)
        obj = (logit * g).sum() + ((0.5 * (s.square() - E.square().sum(1)).sum(1)) * g).sum()
    obj.backward()

    c = lambda t: t.detach().float().cuda().contiguous()
    assert ops.mlp_supported(FD, Dn, list(hidden))
    h_out = [torch.full((B, 32), 7.0, device="cuda") for _ in hidden]
    out = torch.empty(B, device="cuda")
    ops.mlp_fwd(c(xe), c(xd) if Dn else None, [c(W) for W in Ws], [c(b) for b in bs], c(w_out), c(w0),
                act, h_out, out)
    torch.cuda.synchronize()

    def close(got, want, what, tol=2e-5):
        want = want.double()
        scale = max(1.0, float(want.abs().max()))
        err = float((got.cpu().double() - want).abs().max())
        assert err <= tol * scale, f"{what}: {err:.3e} (scale {scale:.3e})"

    close(out, logit, "logit")
    for l, hl in enumerate(hs):
        close(h_out[l][:, : hidden[l]], hl, f"h{l}")
        assert float(h_out[l][:, hidden[l]:].abs().max()) == 0.0 if hidden[l] < 32 else True

    d_rows = torch.full((B, FD), 3.0, device="cuda")
    dh = [torch.empty(B, 32, device="cuda") for _ in hidden]
    dW = [torch.empty_like(c(W)) for W in Ws]
    ws = torch.empty(ops.mlp_bwd_workspace(FD, Dn), device="cuda")
    db = [torch.empty(hh, device="cuda") for hh in hidden]
    dwo, dw0 = torch.empty(hidden[-1], device="cuda"), torch.empty(1, device="cuda")
    ops.mlp_bwd(c(xe), c(xd) if Dn else None, [c(W) for W in Ws], c(w_out), act, c(g), h_out, d_rows,
                dh, dW, ws, fm_sum=c(S) if fm else None, db=db, d_w_out=dwo, d_w0_out=dw0)
    torch.cuda.synchronize()
    close(d_rows, xe.grad, "d_rows")
    for l in range(len(hidden)):
        close(dW[l], Ws[l].grad, f"dW{l}")
        close(dh[l][:, : hidden[l]].sum(0), bs[l].grad, f"db{l} (colsum dh)")
        close(db[l], bs[l].grad, f"db{l}")
    close(dwo, w_out.grad, "d w_out")
    close(dw0, g.sum().reshape(1), "d w0")


def test_mlp_unsupported_shapes_reported(hip_lib):
    from recman_amd import ops

    assert not ops.mlp_supported(416, 13, [400, 400])
    assert not ops.mlp_supported(416, 13, [32, 32, 32, 32])
    assert not ops.mlp_supported(512, 0, [32])
    assert ops.mlp_supported(416, 13, [32, 32])
