"""rm_deepfm_step (DeepFM's forward and EVERY gradient in one kernel) against the CPU oracle and against the
two-kernel path (rm_embed_mlp_fwd + rm_mlp_bwd) it replaces in the DeepFM engine.  The one-kernel step sums
layer 0 over the fields in another order (7 per-worker partial sums) and uses the 16x16x4 MFMA, so it is held
to the parity tolerances, not to bit equality, against the two-kernel path."""
import pytest
import torch

from oracle import th_layers as T
from tests.cases import make_case
from tests.test_gpu_parity import _close, _close_grad, _engine

pytestmark = pytest.mark.gpu

CASES = [
    dict(B=37, F=5, Dn=3, hidden=(32, 32)),
    dict(B=300, F=26, Dn=13, hidden=(32, 32), scale=0.05),   # the Criteo shape
    dict(B=64, F=4, Dn=0, hidden=(16, 8)),                   # no dense inputs, narrow layers, full tiles
    dict(B=33, F=7, Dn=16, hidden=(32, 24)),                 # Dn at its limit, ragged last tile
    dict(B=16 * 9 + 1, F=26, Dn=16, hidden=(8, 8)),          # the widest x the kernel takes
    dict(B=1, F=1, Dn=1, hidden=(32, 1)),
    dict(B=1000, F=21, Dn=2, hidden=(24, 32), scale=0.1),    # several tiles per block round, 3-slot workers
]


def _id(c):
    return f"B{c['B']}F{c['F']}Dn{c['Dn']}H{'x'.join(map(str, c['hidden']))}"


@pytest.mark.parametrize("case", CASES, ids=_id)
@pytest.mark.parametrize("act", ["relu", "leaky_relu"])
def test_one_kernel_step_matches_the_oracle_and_the_two_kernel_path(hip_lib, case, act):
    spec, p, idx, dense, y, hp = make_case("deepfm", D=16, **case)
    hp = dict(hp, deep_activation=act)
    loss_o, logit_o, pred_o, grads_o = T.fwd_bwd("deepfm", p, spec, idx, dense, y, hp)
    e1 = _engine("deepfm", spec, 16, dict(hp, step_fusion=True), p)
    e2 = _engine("deepfm", spec, 16, dict(hp, step_fusion=False), p)
    idx_d, dense_d, y_d = idx.cuda(), dense.cuda(), y.cuda()
    l1 = e1.fwd_bwd(idx_d, dense_d, y_d)
    l2 = e2.fwd_bwd(idx_d, dense_d, y_d)
    torch.cuda.synchronize()
    assert e1._step_ok is True and getattr(e2, "_step_ok", None) is None
    _close(e1.logit, logit_o, rtol=0, atol=1e-5, what="logit vs oracle")
    _close(e1.pred, pred_o, rtol=0, atol=1e-5, what="pred vs oracle")
    _close(l1, loss_o.reshape(1), what="loss vs oracle")
    _close(e1.logit, e2.logit, rtol=1e-6, atol=1e-6, what="logit vs two kernels")  # (other summation order over the fields)
    _close(l1, l2, rtol=1e-6, what="loss vs two kernels")
    _close_grad(e1.dlogit, e2.dlogit, what="dlogit vs two kernels")
    _close_grad(e1.d_rows, e2.d_rows, what="row gradients vs two kernels")
    g1 = e1.dense_grads(idx_d, reference_names=True)
    for k in g1:
        _close_grad(g1[k], grads_o[k], what=f"grad {k}")


def test_one_kernel_step_float_labels_regression_and_grad_scale(hip_lib):
    spec, p, idx, dense, y, hp = make_case("deepfm", D=16, B=130, F=9, Dn=4, hidden=(32, 16))
    yf = torch.randn(130)
    loss_o, logit_o, _, grads_o = T.fwd_bwd("deepfm", p, spec, idx, dense, yf, hp, task="regression")
    from recman_amd import engine as eng

    e = eng.DeepFMEngine(eng.FeatureSpec(spec.sparse_names, spec.feat_sizes, spec.dense_names), 16,
                         dict(hp, step_fusion=True),
                         task="regression", device="cuda:0")
    e.load_params({k: v for k, v in p.items() if k in e.params or k == "linear_w"})
    idx_d, dense_d = idx.cuda(), dense.cuda()
    l = e.fwd_bwd(idx_d, dense_d, yf.cuda())
    assert e._step_ok is True
    _close(e.logit, logit_o, rtol=0, atol=1e-5, what="logit")
    _close(l, loss_o.reshape(1), what="loss")
    g = e.dense_grads(idx_d, reference_names=True)
    for k in g:
        _close_grad(g[k], grads_o[k], what=f"grad {k}")
    # grad_scale (a micro-batch's share of the step): every gradient scales, the loss does not
    d0, w0 = e.d_rows.clone(), e.grads["dnn_layer_0_weights"].clone()
    e.grad_scale = 0.25
    e.fwd_bwd(idx_d, dense_d, yf.cuda())
    reg = hp["deep_l2_reg"]
    _close_grad(e.d_rows, 0.25 * d0, what="scaled row gradients")
    _close_grad(e.grads["dnn_layer_0_weights"] - reg * e.params["dnn_layer_0_weights"],
                0.25 * (w0 - reg * e.params["dnn_layer_0_weights"]), what="scaled dW0")


def test_one_kernel_step_is_deterministic_and_declines_what_it_does_not_cover(hip_lib):
    from recman_amd import ops

    spec, p, idx, dense, y, hp = make_case("deepfm", D=16, B=4099, F=26, Dn=13, hidden=(32, 32), scale=0.05)
    e = _engine("deepfm", spec, 16, dict(hp, step_fusion=True), p)
    idx_d, dense_d, y_d = idx.cuda(), dense.cuda(), y.cuda()
    e.fwd_bwd(idx_d, dense_d, y_d)
    a = [x.clone() for x in (e.logit, e.dlogit, e.d_rows, e.loss, *e.grads.values())]
    e.fwd_bwd(idx_d, dense_d, y_d)
    assert all(torch.equal(x, z) for x, z in zip(a, (e.logit, e.dlogit, e.d_rows, e.loss, *e.grads.values())))
    assert ops.deepfm_step_supported(26, 16, 32, 13, (32, 32))
    assert not ops.deepfm_step_supported(26, 8, 16, 13, (32, 32))     # D = 8 rows
    assert not ops.deepfm_step_supported(27, 16, 32, 13, (32, 32))    # 27 fields + the dense slot > 27 slots
    assert not ops.deepfm_step_supported(26, 16, 32, 17, (32, 32))    # Dn > 16
    assert not ops.deepfm_step_supported(26, 16, 32, 13, (32,))       # one hidden layer
    assert not ops.deepfm_step_supported(26, 16, 32, 13, (64, 32))    # wide hidden layer
    # masks (dropout) keep the separate kernels
    spec2, p2, idx2, dense2, y2, hp2 = make_case("deepfm", D=16, B=40, F=5, Dn=3, hidden=(32, 32))
    e2 = _engine("deepfm", spec2, 16, dict(hp2, use_fm=False, step_fusion=True), p2)
    e2.fwd_bwd(idx2.cuda(), dense2.cuda(), y2.cuda())
    assert getattr(e2, "_step_ok", None) is None   # FM off: not this kernel's model
