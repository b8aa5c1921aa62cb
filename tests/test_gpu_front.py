"""rm_embed_mlp_fwd (gather + FM + linear + skinny MLP + head in one kernel) against the two-kernel path
(rm_embed_fwd + rm_mlp_fwd) it replaces in the DeepFM engine, and against the CPU oracle.  The gathered
values and FM sums follow the same arithmetic in the same order: E and fm_sum must be BIT-identical, lin_logit
and fm_logit agree to 1e-6 (the linear term's dense part and the sum of squares are added in another order); the MLP's layer 0 accumulates in the same k order per accumulator, so the
logits are held to 1e-6 and the gradients to the parity tolerance."""
import pytest
import torch

from oracle import th_layers as T
from tests.cases import make_case
from tests.test_gpu_parity import _close, _close_grad, _engine

pytestmark = pytest.mark.gpu

CASES = [
    dict(B=37, F=5, Dn=3, hidden=(32, 32)),
    dict(B=300, F=26, Dn=13, hidden=(32, 32), scale=0.05),   # the Criteo shape
    dict(B=64, F=4, Dn=0, hidden=(16,)),                     # no dense inputs, one layer, full chunk
    dict(B=33, F=7, Dn=16, hidden=(32, 24, 8)),              # three layers, Dn at its limit, ragged tile
    dict(B=129, F=27, Dn=16, hidden=(8, 8)),                 # K = 448: the widest x the kernel takes
    dict(B=1, F=1, Dn=1, hidden=(32,)),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"B{c['B']}F{c['F']}Dn{c['Dn']}H{len(c['hidden'])}")
@pytest.mark.parametrize("use_fm", [True, False])
def test_one_kernel_front_equals_the_two_kernel_path_and_the_oracle(hip_lib, case, use_fm):
    spec, p, idx, dense, y, hp = make_case("deepfm", D=16, **case)
    hp = dict(hp, use_fm=use_fm)
    loss_o, logit_o, pred_o, grads_o = T.fwd_bwd("deepfm", p, spec, idx, dense, y, hp)
    e1 = _engine("deepfm", spec, 16, dict(hp, front_fusion=True, step_fusion=False), p)   # (the one-kernel STEP has its own test)
    e2 = _engine("deepfm", spec, 16, dict(hp, front_fusion=False), p)
    idx_d, dense_d, y_d = idx.cuda(), dense.cuda(), y.cuda()
    l1 = e1.fwd_bwd(idx_d, dense_d, y_d)
    l2 = e2.fwd_bwd(idx_d, dense_d, y_d)
    torch.cuda.synchronize()
    assert e1._front_ok is True and getattr(e2, "_front_ok", None) is None
    assert torch.equal(e1.E, e2.E)
    _close(e1.lin_logit, e2.lin_logit, rtol=0, atol=1e-6, what="lin_logit vs two kernels")  # (dense part summed in another order)
    if use_fm:
        assert torch.equal(e1.fm_sum, e2.fm_sum)
        _close(e1.fm_logit, e2.fm_logit, rtol=0, atol=1e-6, what="fm_logit vs two kernels")  # (sum of squares: contraction)
    _close(e1.logit, e2.logit, rtol=0, atol=1e-6, what="logit vs two kernels")
    _close(l1, l2, rtol=1e-6, what="loss vs two kernels")
    _close(e1.logit, logit_o, rtol=0, atol=1e-5, what="logit vs oracle")
    _close(l1, loss_o.reshape(1), what="loss vs oracle")
    g1 = e1.dense_grads(idx_d, reference_names=True)
    for k in g1:
        _close_grad(g1[k], grads_o[k], what=f"grad {k}")
    _close_grad(e1.d_rows, e2.d_rows, what="row gradients vs two kernels")
    # inference (no head)
    li, _ = e1.forward(idx_d, dense_d, training=False)
    _close(li, logit_o, rtol=0, atol=1e-5, what="inference logit")


def test_one_kernel_front_is_deterministic_and_declines_what_it_does_not_cover(hip_lib):
    from recman_amd import ops

    spec, p, idx, dense, y, hp = make_case("deepfm", D=16, B=200, F=26, Dn=13, hidden=(32, 32))
    e = _engine("deepfm", spec, 16, dict(hp, step_fusion=False), p)
    idx_d, dense_d, y_d = idx.cuda(), dense.cuda(), y.cuda()
    e.fwd_bwd(idx_d, dense_d, y_d)
    a = (e.logit.clone(), e.dlogit.clone(), e.d_rows.clone())
    e.fwd_bwd(idx_d, dense_d, y_d)
    assert all(torch.equal(x, z) for x, z in zip(a, (e.logit, e.dlogit, e.d_rows)))
    assert ops.embed_mlp_fwd_supported(26, 16, 32, 13, (32, 32))
    assert not ops.embed_mlp_fwd_supported(26, 8, 16, 13, (32, 32))    # D = 8 rows
    assert not ops.embed_mlp_fwd_supported(26, 16, 32, 17, (32, 32))   # Dn > 16
    assert not ops.embed_mlp_fwd_supported(28, 16, 32, 13, (32, 32))   # K > 448
    assert not ops.embed_mlp_fwd_supported(26, 16, 32, 13, (64, 32))   # wide hidden layer
    # D = 8 engines keep the two-kernel path
    spec8, p8, idx8, dense8, y8, hp8 = make_case("deepfm", D=8, B=50)
    e8 = _engine("deepfm", spec8, 8, hp8, p8)
    e8.fwd_bwd(idx8.cuda(), dense8.cuda(), y8.cuda())
    assert e8._front_ok is False
