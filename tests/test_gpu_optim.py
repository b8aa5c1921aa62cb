"""GPU: the row-wise sparse optimizer step against the dense Keras-semantics optimizer
(they coincide when the optimizer is rebuilt every batch, the reference's behaviour) and
against a lazy-Adam restatement in torch for the persistent case."""
import math

import numpy as np
import pytest
import torch

from tests.cases import make_case

pytestmark = pytest.mark.gpu


def _engine(model, spec, D, hp, p):
    from recman_amd import engine as eng

    e = eng.ENGINES[model](eng.FeatureSpec(spec.sparse_names, spec.feat_sizes, spec.dense_names), D, hp)
    e.load_params({k: v for k, v in p.items() if k in e.params or k == "linear_w"})
    return e


@pytest.mark.parametrize("name", ["adam", "adagrad", "sgd"])
def test_sparse_step_equals_dense_step_when_reset_every_batch(hip_lib, name):
    from recman_amd.optim import Optimizer, SparseTableOptimizer

    spec, p, idx, dense, y, hp = make_case("deepfm", B=300, D=16, sizes=[7, 11, 5, 13, 3])
    hp = dict(hp, embedding_l2_reg=0.0, linear_l2_reg=0.0)
    e1, e2 = _engine("deepfm", spec, 16, hp, p), _engine("deepfm", spec, 16, hp, p)
    dopt = Optimizer(name, 0.01)
    sopt, sdense = SparseTableOptimizer(e2, name, 0.01), Optimizer(name, 0.01)
    idx_d, dense_d, y_d = idx.cuda(), dense.cuda(), y.cuda()
    for step in range(3):
        e1.fwd_bwd(idx_d, dense_d, y_d)
        dopt.reset()
        dopt.step(e1.params, e1.dense_grads(idx_d))
        e2.fwd_bwd(idx_d, dense_d, y_d)
        sdense.reset()
        sopt.step(idx_d, reset=True)
        sdense.step(e2.params, e2.grads)
        for k in e1.params:
            a, b = e1.params[k], e2.params[k]
            assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max())), (step, k)
    assert float(sopt.gbuf.abs().max()) == 0.0  # the gradient buffer is clean again


def test_lazy_adam_persistent_state_matches_torch_restatement(hip_lib):
    from recman_amd.optim import SparseTableOptimizer

    spec, p, idx, dense, y, hp = make_case("xdeepfm", B=120, D=8, cin_units=(8, 4), scale=0.2)
    hp = dict(hp, embedding_l2_reg=0.0, linear_l2_reg=0.0)
    e = _engine("xdeepfm", spec, 8, hp, p)
    sopt = SparseTableOptimizer(e, "adam", 0.01)
    R, LD = e.rows.shape
    ref = e.rows.detach().clone().double()
    m, v = torch.zeros_like(ref), torch.zeros_like(ref)
    g = torch.Generator().manual_seed(3)
    for t in range(1, 4):
        bi = torch.randint(0, 120, (60,), generator=g)
        ib, db, yb = idx[bi].cuda().contiguous(), dense[bi].cuda().contiguous(), y[bi].cuda().contiguous()
        e.fwd_bwd(ib, db, yb)
        # touched rows and their summed gradients (xDeepFM: no bias tables, linear term on)
        rows_g = (ib + e.field_off).reshape(-1)
        G = torch.zeros(R, LD, dtype=torch.float64, device="cuda")
        G[:, :8].index_add_(0, rows_g, e.d_rows.reshape(-1, 8).double())
        G[:, 9].index_add_(0, rows_g, e.dlogit.double().repeat_interleave(e.F))
        touched = torch.zeros(R, dtype=torch.bool, device="cuda")
        touched[rows_g] = True
        mt = 0.9 * m + 0.1 * G
        vt = 0.999 * v + 0.001 * G * G
        lr_t = 0.01 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        upd = ref - lr_t * mt / (vt.sqrt() + 1e-7)
        tm = touched[:, None]
        m, v, ref = torch.where(tm, mt, m), torch.where(tm, vt, v), torch.where(tm, upd, ref)
        sopt.step(ib)
        cols = list(range(8)) + [9]
        assert float((e.rows.double()[:, cols] - ref[:, cols]).abs().max()) < 1e-6, t


def test_model_fit_with_sparse_optimizer_learns(hip_lib):
    import pandas as pd
    from sklearn.metrics import log_loss

    import recman_amd.th as th
    from tests.test_gpu_models import ml_features, ml_frame

    df = ml_frame()
    fd = ml_features(df)
    hp = {"embedding_size": 8, "deep_dropout": (1, 1, 1), "cin_cross_layer_units": [16, 16],
          "cin_dropout": [1, 1, 1], "learning_rate": 0.01, "embedding_l2_reg": 0.0, "linear_l2_reg": 0.0,
          "sparse_optimizer": True}
    m = th.xDeepFM(fd, hp, epoch=3, batch_size=128)
    before = log_loss(df["label"].values, m.predict(df).astype(np.float64))
    m.fit(df, df["label"].values)
    assert m._sparse_opt is not None and m._sparse_opt.t > 0
    after = log_loss(df["label"].values, m.predict(df).astype(np.float64))
    assert after < before - 0.01


def test_sparse_step_with_multi_valued_and_value_features(hip_lib):
    """Scratch-row features in the row-wise step: a MultiValCsvFeat and a SparseValueFeat field are
    masked out of the main call and arrive as expanded one-field occurrence lists.  With the
    optimizer rebuilt per batch (reset) the result equals the dense Adam step on dense_grads()."""
    from oracle import th_layers as T
    from recman_amd import engine as eng
    from recman_amd.optim import Optimizer, SparseTableOptimizer

    spec, p, idx, dense, y, hp = make_case("deepfm", B=90, D=8, sizes=[7, 11, 5, 13, 3])
    hp = dict(hp, embedding_l2_reg=0.0, linear_l2_reg=0.0)
    mname, vname = spec.sparse_names[1], spec.sparse_names[3]
    tspec = T.Spec(spec.sparse_names, spec.feat_sizes, spec.dense_names, multi_names=[mname],
                   value_names=[vname])
    espec = eng.FeatureSpec(spec.sparse_names, spec.feat_sizes, spec.dense_names, [mname], [vname])
    g = torch.Generator().manual_seed(17)
    B = 90
    n = torch.randint(0, 4, (B,), generator=g)
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64), n.cumsum(0)])
    ids = torch.randint(0, spec.feat_sizes[1], (int(n.sum()),), generator=g)
    vids = torch.randint(0, spec.feat_sizes[3], (B,), generator=g)
    vals = torch.randn(B, generator=g)
    mv = {mname: (offsets.cuda(), ids.cuda()), vname: (torch.arange(B + 1).cuda(), vids.cuda(), vals.cuda())}

    def mk():
        e = eng.ENGINES["deepfm"](espec, 8, hp)
        e.load_params({k: v for k, v in p.items() if k in e.params or k == "linear_w"})
        return e

    e1, e2 = mk(), mk()
    dopt, sdense = Optimizer("adam", 0.01), Optimizer("adam", 0.01)
    sopt = SparseTableOptimizer(e2, "adam", 0.01)
    idx_d, dense_d, y_d = idx.cuda(), dense.cuda(), y.cuda()
    for step in range(2):
        e1.fwd_bwd(idx_d, dense_d, y_d, mv=mv)
        dopt.reset()
        dopt.step(e1.params, e1.dense_grads(idx_d))
        e2.fwd_bwd(idx_d, dense_d, y_d, mv=mv)
        sdense.reset()
        sopt.step(idx_d, reset=True)
        sdense.step(e2.params, e2.grads)
        for k in e1.params:
            a, b = e1.params[k], e2.params[k]
            assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max())), (step, k)
    assert float(sopt.gbuf.abs().max()) == 0.0
    assert tspec.multi_names == [mname]  # (the oracle spec of the same case, for readers)
