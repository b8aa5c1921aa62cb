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
    assert tspec.multi_names == [mname]  # (the oracle spec of the same case, for readers)


def _run_steps(e, sopt, dense_opt, batches, reset=False):
    for ib, db, yb in batches:
        e.fwd_bwd(ib, db, yb)
        sopt.step(ib, reset=reset)
        dense_opt.step(e.params, e.grads)


@pytest.mark.parametrize("name", ["adam", "adagrad"])
def test_lazy_equals_keras_dense_when_every_row_is_touched_and_diverges_otherwise(hip_lib, name):
    """LazyAdam (the row-wise step) vs Keras' sparse apply, which decays the moments of EVERY row
    (Optimizer on the densified gradient does exactly that): over several steps with PERSISTENT state
    the two coincide when every table row occurs in every batch, and differ for rows a batch leaves
    out - the documented deviation (csrc/optim.hip)."""
    from recman_amd.optim import Optimizer, SparseTableOptimizer

    sizes = [3, 4, 2]
    spec, p, idx, dense, y, hp = make_case("deepfm", B=200, F=3, D=8, sizes=sizes)
    hp = dict(hp, embedding_l2_reg=0.0, linear_l2_reg=0.0)
    g = torch.Generator().manual_seed(5)
    # batches that contain every row of every field (200 draws over <= 4 rows per field)
    full = []
    for _ in range(4):
        ib = torch.stack([torch.randint(0, v, (200,), generator=g) for v in sizes], 1)
        for f, v in enumerate(sizes):
            ib[:v, f] = torch.arange(v)
        full.append((ib.cuda(), dense.cuda(), y.cuda()))
    e1, e2 = _engine("deepfm", spec, 8, hp, p), _engine("deepfm", spec, 8, hp, p)
    dopt, sdense = Optimizer(name, 0.01), Optimizer(name, 0.01)
    sopt = SparseTableOptimizer(e2, name, 0.01)
    for ib, db, yb in full:
        e1.fwd_bwd(ib, db, yb)
        dopt.step(e1.params, e1.dense_grads(ib))
    _run_steps(e2, sopt, sdense, full)
    for k in e1.params:
        a, b = e1.params[k], e2.params[k]
        assert float((a - b).abs().max()) <= 5e-6 * max(1.0, float(a.abs().max())), k
    # one more step that leaves row 0 of field 0 out: Keras still moves it (its m is non-zero), lazy does not
    ib = full[0][0].clone()
    ib[:, 0] = ib[:, 0].clamp(min=1)
    row0_before = e2.params[f"{spec.sparse_names[0]}_feat_embed"][0].clone()
    e1.fwd_bwd(ib, full[0][1], full[0][2])
    dopt.step(e1.params, e1.dense_grads(ib))
    _run_steps(e2, sopt, sdense, [(ib, full[0][1], full[0][2])])
    k = f"{spec.sparse_names[0]}_feat_embed"
    assert torch.equal(e2.params[k][0], row0_before)                      # lazy: untouched row unchanged
    if name == "adam":
        assert float((e1.params[k][0] - row0_before).abs().max()) > 1e-5  # Keras: it moved
    else:  # Adagrad has no momentum: a zero gradient moves nothing in either form
        assert float((e1.params[k][0] - row0_before).abs().max()) < 1e-7
    assert float((e1.params[k][1:] - e2.params[k][1:]).abs().max()) <= 5e-6  # touched rows still agree


@pytest.mark.parametrize("name", ["adam", "gd"])
def test_lazy_l2_equals_the_dense_l2_term_when_every_row_is_touched_and_diverges_otherwise(hip_lib, name):
    """embedding_l2_reg / linear_l2_reg (FeatEmbedding.l2 / LinearLayer.l2, layers.py:188-193, 349-354) applied
    LAZILY by the row-wise step - reg * row for the rows a batch touches - against the reference's dense term
    (the densified gradient + reg * table, through Optimizer): identical while every row occurs in every batch,
    and a row a batch leaves out keeps its value where the dense term would shrink it."""
    from recman_amd.optim import Optimizer, SparseTableOptimizer

    sizes = [3, 4, 2]
    spec, p, idx, dense, y, hp = make_case("deepfm", B=200, F=3, D=8, sizes=sizes)
    reg_e, reg_l = 3e-2, 2e-2
    hp = dict(hp, embedding_l2_reg=reg_e, linear_l2_reg=reg_l, deep_l2_reg=0.0)
    g = torch.Generator().manual_seed(7)
    full = []
    for _ in range(3):
        ib = torch.stack([torch.randint(0, v, (200,), generator=g) for v in sizes], 1)
        for f, v in enumerate(sizes):
            ib[:v, f] = torch.arange(v)
        full.append((ib.cuda(), dense.cuda(), y.cuda()))
    e1 = _engine("deepfm", spec, 8, hp, p)                        # dense l2 (the reference's term)
    e2 = _engine("deepfm", spec, 8, dict(hp, lazy_l2=True), p)    # lazy
    dopt, sdense = Optimizer(name, 0.01), Optimizer(name, 0.01)
    sopt = SparseTableOptimizer(e2, name, 0.01, l2_embedding=reg_e, l2_linear=reg_l)
    for ib, db, yb in full:
        e1.fwd_bwd(ib, db, yb)
        dopt.step(e1.params, e1.dense_grads(ib))
    _run_steps(e2, sopt, sdense, full)
    for k in e1.params:
        a, b = e1.params[k], e2.params[k]
        assert float((a - b).abs().max()) <= 5e-6 * max(1.0, float(a.abs().max())), k
    # a step without row 0 of field 0: the dense term still shrinks it, the lazy one leaves it alone
    ib = full[0][0].clone()
    ib[:, 0] = ib[:, 0].clamp(min=1)
    k = f"{spec.sparse_names[0]}_feat_embed"
    row0 = e2.params[k][0].clone()
    e1.fwd_bwd(ib, full[0][1], full[0][2])
    dopt.step(e1.params, e1.dense_grads(ib))
    _run_steps(e2, sopt, sdense, [(ib, full[0][1], full[0][2])])
    assert torch.equal(e2.params[k][0], row0)
    assert float((e1.params[k][0] - row0).abs().max()) > 1e-6
    assert float((e1.params[k][1:] - e2.params[k][1:]).abs().max()) <= 5e-6
    # the FM bias entries carry no l2 in either form (layers.py:188-193 sums the embedding tables only)
    kb = f"{spec.sparse_names[1]}_feat_bias"
    assert float((e1.params[kb] - e2.params[kb]).abs().max()) <= 5e-6


def test_sparse_step_is_bit_reproducible_with_heavy_duplicates(hip_lib):
    """Runs of equal rows far beyond the short-run kernel's cap (a 3-row field over 4,000 examples): the
    one-wave-per-run kernel sums in a fixed order; two runs from the same state are bit-identical and
    equal a float64 restatement to fp32 rounding."""
    from recman_amd import ops

    B, F, D, sizes = 4000, 3, 16, [3, 50, 100000]
    g = torch.Generator().manual_seed(11)
    idx = torch.stack([torch.randint(0, v, (B,), generator=g) for v in sizes], 1).cuda()
    idx[::7, 1] = -1                                   # skipped occurrences
    foff = torch.tensor([0, sizes[0], sizes[0] + sizes[1]]).cuda()
    R, LD = sum(sizes), 2 * D
    d_rows = torch.randn(B, F, D, generator=g).cuda()
    gb, gl = torch.randn(B, generator=g).cuda(), torch.randn(B, generator=g).cuda()
    rows0 = torch.randn(R, LD, generator=g).cuda()
    rows0[:, D + 2: D + 6] = 0
    ws = torch.zeros(ops.sparse_optimizer_workspace(B * F), dtype=torch.uint8, device="cuda")
    outs = []
    for _ in range(2):
        rows, mom = rows0.clone(), torch.zeros(R, 2 * D, device="cuda")
        for t in (1, 2):
            ops.sparse_optimizer_step(idx, foff, d_rows, rows, mom, ws, t, "adam", 0.01, g_bias=gb, g_lin=gl)
        outs.append((rows, mom))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # float64 restatement of two lazy Adam steps with the same gradients
    ok = (idx >= 0)
    rg = (idx + foff).clamp(min=0)[ok]
    G = torch.zeros(R, D + 2, dtype=torch.float64, device="cuda")
    G[:, :D].index_add_(0, rg, d_rows[ok].double())
    G[:, D].index_add_(0, rg, gb.double()[:, None].expand(B, F)[ok])
    G[:, D + 1].index_add_(0, rg, gl.double()[:, None].expand(B, F)[ok])
    touched = torch.zeros(R, dtype=torch.bool, device="cuda")
    touched[rg] = True
    P = rows0[:, : D + 2].double()
    m = torch.zeros_like(P)
    v = torch.zeros_like(P)
    for t in (1, 2):
        m = 0.9 * m + 0.1 * G
        v = 0.999 * v + 0.001 * G * G
        P = P - 0.01 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (v.sqrt() + 1e-7)
    want = torch.where(touched[:, None], P, rows0[:, : D + 2].double())
    got = outs[0][0][:, : D + 2].double()
    assert float((got - want).abs().max()) < 5e-6
    assert torch.equal(outs[0][0][~touched], rows0[~touched])  # untouched rows: not a bit changes


def test_step_rows_entry_equals_the_indexed_slices_entry(hip_lib):
    """rm_sparse_optimizer_step_rows (the row-sharded table's owner side: gradient rows that carry their
    local row, [dE | g_bias | g_lin | pad]) gives bit-identical results to rm_sparse_optimizer_step on
    the same occurrences."""
    from recman_amd import ops

    B, F, D = 500, 4, 8
    sizes = [5, 40, 300, 7]
    g = torch.Generator().manual_seed(2)
    idx = torch.stack([torch.randint(0, v, (B,), generator=g) for v in sizes], 1).cuda()
    foff = torch.tensor([0, 5, 45, 345]).cuda()
    R, LD = sum(sizes), 16
    d_rows = torch.randn(B, F, D, generator=g).cuda()
    gb, gl = torch.randn(B, generator=g).cuda(), torch.randn(B, generator=g).cuda()
    rows0 = torch.randn(R, LD, generator=g).cuda()
    rows0[:, D + 2: D + 6] = 0
    ws = torch.zeros(ops.sparse_optimizer_workspace(B * F), dtype=torch.uint8, device="cuda")
    for kind in ("adam", "adagrad", "sgd"):
        rows_a, rows_b = rows0.clone(), rows0.clone()
        mom_a = None if kind == "sgd" else torch.zeros(R, 2 * D, device="cuda")
        mom_b = None if kind == "sgd" else torch.zeros(R, 2 * D, device="cuda")
        packed = torch.zeros(B * F, D + 4, device="cuda")
        packed[:, :D] = d_rows.view(-1, D)
        packed[:, D] = gb.repeat_interleave(F)
        packed[:, D + 1] = gl.repeat_interleave(F)
        ids = (idx + foff).reshape(-1).contiguous()
        for t in (1, 2, 3):
            ops.sparse_optimizer_step(idx, foff, d_rows, rows_a, mom_a, ws, t, kind, 0.05, g_bias=gb, g_lin=gl)
            ops.sparse_optimizer_step_rows(ids, packed, D, rows_b, mom_b, ws, t, kind, 0.05)
        assert torch.equal(rows_a, rows_b), kind
        if mom_a is not None:
            assert torch.equal(mom_a, mom_b), kind
        assert not torch.equal(rows_a, rows0)


@pytest.mark.parametrize("name", ["adam", "adagrad", "sgd"])
def test_fused_dense_optimizer_equals_the_per_tensor_one(hip_lib, name):
    from recman_amd.optim import FusedDenseOptimizer, Optimizer

    spec, p, idx, dense, y, hp = make_case("xdeepfm", B=64, D=8, cin_units=(8, 4), scale=0.2)
    e1, e2 = _engine("xdeepfm", spec, 8, hp, p), _engine("xdeepfm", spec, 8, hp, p)
    a, b = Optimizer(name, 0.01), FusedDenseOptimizer(e2, name, 0.01)
    assert e2.params["linear_w_dense"].data_ptr() == e2.linear_w_dense.data_ptr()
    ib, db, yb = idx.cuda(), dense.cuda(), y.cuda()
    for step in range(3):
        e1.fwd_bwd(ib, db, yb)
        a.step(e1.params, e1.grads)
        e2.fwd_bwd(ib, db, yb)
        b.step()
        for k in e1.grads:
            assert float((e1.params[k] - e2.params[k]).abs().max()) <= 2e-6 * max(1.0, float(e1.params[k].abs().max())), (step, k)
    assert float((e1.logit - e2.logit).abs().max()) < 1e-5


def test_prepared_step_equals_the_one_call_step(hip_lib):
    """rm_sparse_optimizer_prepare + step(prepared) == step: fit() sorts the ids on a side stream while
    the forward+backward pass runs."""
    from recman_amd.optim import SparseTableOptimizer

    spec, p, idx, dense, y, hp = make_case("deepfm", B=257, D=16)
    hp = dict(hp, embedding_l2_reg=0.0, linear_l2_reg=0.0)
    e1, e2 = _engine("deepfm", spec, 16, hp, p), _engine("deepfm", spec, 16, hp, p)
    s1, s2 = SparseTableOptimizer(e1, "adam", 0.01), SparseTableOptimizer(e2, "adam", 0.01)
    ib, db, yb = idx.cuda(), dense.cuda(), y.cuda()
    side = torch.cuda.Stream()
    for _ in range(3):
        e1.fwd_bwd(ib, db, yb)
        s1.step(ib)
        s2._workspace(ib.numel())
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            s2.prepare(ib)
        e2.fwd_bwd(ib, db, yb)
        torch.cuda.current_stream().wait_stream(side)
        assert s2._prepared is not None
        s2.step(ib)
        assert s2._prepared is None
        assert torch.equal(e1.rows, e2.rows) and torch.equal(s1.mom, s2.mom)
    m, v = s1.moments()
    assert m.shape == (e1.rows.shape[0], 16) and float(v.min()) >= 0 and float(m.abs().max()) > 0


@pytest.mark.parametrize("case", [
    dict(B=4000, sizes=[3, 50, 100000]),                        # long runs, two passes, ragged last sub-block
    dict(B=1024, sizes=[1]),                                    # one row: everything is one run
    dict(B=2500, sizes=[(1 << 20) - 1, 17, 1 << 20, 5]),        # 2^20 rows: the 21st bit -> three passes
    dict(B=1, sizes=[9, 9]),
    dict(B=70000, sizes=[1000001] * 5 + [40]),                  # the Criteo field size, > 64 sub-blocks per field
    dict(B=300, sizes=[2] * 64),                                # the field limit
], ids=lambda c: f"B{c['B']}F{len(c['sizes'])}")
def test_field_segmented_sort_equals_the_sort_over_all_pairs(hip_lib, case):
    """max_field_rows > 0 (ids sorted per field on their local bits, csrc/optim.hip) leaves the SAME table and
    moments, bit for bit, as the one stable sort over all (row, occurrence) pairs: the runs of equal rows hold
    the same occurrences in the same order.  Negative ids are skipped by both."""
    from recman_amd import ops

    B, sizes, D = case["B"], case["sizes"], 16
    F = len(sizes)
    g = torch.Generator().manual_seed(5)
    cols = []
    for v in sizes:
        c = torch.randint(0, v, (B,), generator=g)
        if v > 1000:  # heavy duplicates inside a large field as well
            c[: B // 3] = c[: B // 3] % 7
        cols.append(c)
    idx = torch.stack(cols, 1)
    idx[::11, 0] = -1
    if F > 1:
        idx[5::13, F - 1] = -1
    idx = idx.cuda()
    foff = torch.tensor([0] + list(np.cumsum(sizes)[:-1]), dtype=torch.int64).cuda()
    R, LD = int(sum(sizes)), 2 * D
    d_rows = torch.randn(B, F, D, generator=g).cuda()
    gb, gl = torch.randn(B, generator=g).cuda(), torch.randn(B, generator=g).cuda()
    rows0 = torch.randn(R, LD, generator=g).cuda()
    rows0[:, D + 2: D + 6] = 0
    ws = torch.zeros(ops.sparse_optimizer_workspace(B * F), dtype=torch.uint8, device="cuda")
    outs = []
    for mfr in (0, max(sizes)):
        rows, mom = rows0.clone(), torch.zeros(R, 2 * D, device="cuda")
        for t in (1, 2):
            ops.sparse_optimizer_step(idx, foff, d_rows, rows, mom, ws, t, "adam", 0.01, g_bias=gb, g_lin=gl,
                                      max_field_rows=mfr, l2_embedding=1e-3)
        # the prepared form
        ops.sparse_optimizer_prepare(ws, R, idx=idx, field_off=foff, max_field_rows=mfr)
        ops.sparse_optimizer_step(idx, foff, d_rows, rows, mom, ws, 3, "adam", 0.01, g_bias=gb, g_lin=gl,
                                  max_field_rows=mfr, prepared=True)
        torch.cuda.synchronize()
        outs.append((rows, mom))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert not torch.equal(outs[0][0], rows0)


def test_field_segmented_sort_skips_ids_beyond_their_field(hip_lib):
    from recman_amd import ops

    D, sizes, B = 8, [5, 7], 64
    idx = torch.stack([torch.arange(B) % 9, torch.arange(B) % 7], 1).cuda()   # field 0: ids 5..8 are not its rows
    foff = torch.tensor([0, 5]).cuda()
    rows0 = torch.randn(12, 16).cuda()
    rows0[:, D + 2: D + 6] = 0
    d_rows = torch.randn(B, 2, D).cuda()
    ws = torch.zeros(ops.sparse_optimizer_workspace(B * 2), dtype=torch.uint8, device="cuda")
    ok = idx.clone()
    ok[:, 0] = torch.where(ok[:, 0] < 5, ok[:, 0], torch.full_like(ok[:, 0], -1))
    outs = []
    for ids, mfr in ((idx, 7), (ok, 7), (ok, 0)):
        rows = rows0.clone()
        ops.sparse_optimizer_step(ids, foff, d_rows, rows, None, ws, 1, "sgd", 0.1, max_field_rows=mfr)
        outs.append(rows)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
