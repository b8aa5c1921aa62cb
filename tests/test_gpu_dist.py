"""GPU: the row-sharded ENGINE with two ranks - two processes sharing the one GPU of the test box,
collectives over gloo staged through host memory (recman_amd.dist._host_staged).  Everything else is
the product path: HIP routing / gather / pack kernels, fixed-capacity and dynamic exchange layouts,
micro-batch pipelining.  Rank r owns table rows r::2 and its own half of the global batch; the
result must equal the single-GPU engine on the whole batch (gradients of the global-batch mean)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

_PORTS = {}


def _port(base, key):
    """A rendezvous port per test case, derived from the order the cases run in (a `hash()` of the parameters is
    salted per interpreter: collisions would be possible and unreproducible)."""
    return base + _PORTS.setdefault((base, key), sum(1 for k in _PORTS if k[0] == base))


def _worker(rank, world, port, model, kw, fixed, micro, segments, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("RECMAN_FORCE_COLLECTIVES", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys

        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from recman_amd import dist as rd
        from recman_amd import engine as eng
        from tests.cases import make_case

        Bl = 24  # per-rank batch
        spec, p, idx, dense, y, hp = make_case(model, B=world * Bl, D=16, **kw)
        hp = dict(hp, embedding_l2_reg=0.0, linear_l2_reg=0.0, deep_l2_reg=0.0, cross_layer_l2_reg=0.0,
                  cin_l2_reg=0.0)
        espec = eng.FeatureSpec(spec.sparse_names, spec.feat_sizes, spec.dense_names)
        dev = torch.device("cuda", 0)
        s = rd.make_sharded_engine(model, espec, 16, hp, dev, rank, world,
                                   capacity_factor=1.5 if fixed else None, micro_batches=micro)
        s.load_params({k: v for k, v in p.items() if k in s.params})
        full = torch.cat([p[f"{n}_feat_embed"] for n in spec.sparse_names])
        bias = (torch.cat([p[f"{n}_feat_bias"].reshape(-1) for n in spec.sparse_names])
                if model == "deepfm" else None)
        R, D = full.shape[0], 16
        s.st.load_global(full, bias=bias, lin=p["linear_w"].reshape(-1)[:R])
        s.linear_w_dense.copy_(p["linear_w"].reshape(-1)[R:])
        sl = slice(rank * Bl, (rank + 1) * Bl)
        il, dl, yl = idx[sl].cuda(), dense[sl].cuda(), y[sl].cuda()
        if segments:
            # the compute between the collectives replayed from hipGraphs; a second batch through
            # the same graphs (copied into the static inputs), then the batch under test
            s.capture_segments(il.clone(), dl.clone(), yl.clone())  # these become the static inputs
            s.fwd_bwd(il.flip(0).contiguous(), dl.flip(0).contiguous(), yl.flip(0).contiguous())
            assert s._segs is not None
        loss = s.fwd_bwd(il, dl, yl)
        assert not s.overflowed()
        ids, rows = s.shard_grad_ids, s.shard_grad_rows
        if not isinstance(ids, list):
            ids, rows = [ids], [rows]
        dt = torch.zeros(s.st.shard.shape[0], D + rd.PAD, device=dev)
        for i, r in zip(ids, rows):
            live = i >= 0
            dt.index_add_(0, i[live], r[live])
        torch.save({"loss": loss.cpu(), "dt": dt.cpu(), "grads": {k: v.cpu() for k, v in s.grads.items()},
                    "logit": s.logit.cpu()}, f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model,kw,fixed,micro,segments", [
    ("deepfm", {}, True, 1, False),
    ("deepfm", {}, False, 2, False),
    ("dcn", dict(cross_layers=2, scale=0.15), True, 2, False),
    ("xdeepfm", dict(cin_units=(16, 8), scale=0.2), False, 1, False),
    ("deepfm", {}, True, 2, True),
    ("xdeepfm", dict(cin_units=(16, 8), scale=0.2), True, 1, True),
    ("dcn", dict(cross_layers=2, scale=0.15), True, 3, True),
])
def test_two_rank_sharded_engine_equals_single_gpu(hip_lib, tmp_path, model, kw, fixed, micro, segments):
    from recman_amd import dist as rd
    from recman_amd import engine as eng
    from tests.cases import make_case

    world, Bl = 2, 24
    out = str(tmp_path / "r")
    port = _port(29700, repr((model, fixed, micro, segments)))
    mp.spawn(_worker, args=(world, port, model, kw, fixed, micro, segments, out), nprocs=world, join=True)
    res = [torch.load(f"{out}.{r}", weights_only=True) for r in range(world)]

    spec, p, idx, dense, y, hp = make_case(model, B=world * Bl, D=16, **kw)
    hp = dict(hp, embedding_l2_reg=0.0, linear_l2_reg=0.0, deep_l2_reg=0.0, cross_layer_l2_reg=0.0,
              cin_l2_reg=0.0)
    e = eng.ENGINES[model](eng.FeatureSpec(spec.sparse_names, spec.feat_sizes, spec.dense_names), 16, hp)
    e.load_params({k: v for k, v in p.items() if k in e.params or k == "linear_w"})
    loss = e.fwd_bwd(idx.cuda(), dense.cuda(), y.cuda()).cpu()
    gd = e.dense_grads(idx.cuda(), reference_names=True)
    gi = e.dense_grads(idx.cuda())
    R, D = sum(spec.feat_sizes), 16

    def close(got, want, what, atol=2e-6):
        want = want.detach().cpu().double()
        err = float((got.double() - want).abs().max())
        scale = max(1.0, float(want.abs().max()))
        assert err <= atol * scale + 1e-7, f"{what}: {err:.3e}"

    close(torch.stack([r["loss"] for r in res]).mean(0), loss, "loss (mean of the ranks' means)")
    if micro == 1:  # (with micro-batches the engine's logit buffer holds the last micro-batch only)
        close(torch.cat([r["logit"] for r in res]), e.logit, "logit")
    want_t = torch.cat([gd[f"{n}_feat_embed"] for n in spec.sparse_names]).cpu()
    want_l = gd["linear_w"].reshape(-1)[:R].cpu()
    for r in range(world):
        close(res[r]["dt"][:, :D], want_t[r::world], f"table grad of rank {r}'s shard")
        close(res[r]["dt"][:, D + 1], want_l[r::world], f"linear grad of rank {r}'s shard")
        if model == "deepfm":
            want_b = torch.cat([gd[f"{n}_feat_bias"].reshape(-1) for n in spec.sparse_names]).cpu()
            close(res[r]["dt"][:, D], want_b[r::world], f"bias grad of rank {r}'s shard")
        for k, v in res[r]["grads"].items():
            if k in gi:
                close(v, gi[k], f"rank {r} dense grad {k}")


@pytest.mark.parametrize("model,kw,micro", [
    ("deepfm", {}, 2),
    ("dcn", dict(cross_layers=2, scale=0.15), 1),
    ("xdeepfm", dict(cin_units=(16, 8), scale=0.2), 3),
])
def test_segment_graphs_equal_eager_sharded_step(hip_lib, model, kw, micro):
    """World size 1, no collective: the route | gather | compute segments replayed from hipGraphs
    give bit-identical gradients to the same engine launching its kernels eagerly."""
    from recman_amd import dist as rd
    from recman_amd import engine as eng
    from tests.cases import make_case

    B = 24
    spec, p, idx, dense, y, hp = make_case(model, B=B, D=16, **kw)
    hp = dict(hp, embedding_l2_reg=0.0, linear_l2_reg=0.0, deep_l2_reg=0.0, cross_layer_l2_reg=0.0,
              cin_l2_reg=0.0)
    espec = eng.FeatureSpec(spec.sparse_names, spec.feat_sizes, spec.dense_names)
    dev = torch.device("cuda", 0)
    full = torch.cat([p[f"{n}_feat_embed"] for n in spec.sparse_names])
    R = full.shape[0]

    def make():
        s = rd.make_sharded_engine(model, espec, 16, hp, dev, 0, 1, capacity_factor=1.5, micro_batches=micro)
        s.load_params({k: v for k, v in p.items() if k in s.params})
        s.st.load_global(full, lin=p["linear_w"].reshape(-1)[:R])
        s.linear_w_dense.copy_(p["linear_w"].reshape(-1)[R:])
        return s

    def result(s, loss):
        ids, rows = s.shard_grad_ids, s.shard_grad_rows
        if not isinstance(ids, list):
            ids, rows = [ids], [rows]
        dt = torch.zeros(s.st.shard.shape[0], 16 + rd.PAD, device=dev, dtype=torch.float64)
        for i, r in zip(ids, rows):
            live = i >= 0
            dt.index_add_(0, i[live], r[live].double())
        return loss.clone(), dt, {k: v.clone() for k, v in s.grads.items()}

    il, dl, yl = idx.cuda(), dense.cuda(), y.cuda()
    e = make()
    want = result(e, e.fwd_bwd(il, dl, yl))
    s = make()
    s.capture_segments(il.clone(), dl.clone(), yl.clone())
    s.fwd_bwd(il.flip(0).contiguous(), dl.flip(0).contiguous(), yl.flip(0).contiguous())  # another batch
    got = result(s, s.fwd_bwd(il, dl, yl))
    assert not s.overflowed()
    assert torch.equal(got[0], want[0])
    assert float((got[1] - want[1]).abs().max()) <= 1e-6 * max(1.0, float(want[1].abs().max()))
    for k in want[2]:
        assert float((got[2][k] - want[2][k]).abs().max()) <= 1e-6 * max(1.0, float(want[2][k].abs().max())), k


def _train_worker(rank, world, port, model, kw, fixed, micro, segments, l2, lin_names, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("RECMAN_FORCE_COLLECTIVES", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys

        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from recman_amd import dist as rd
        from recman_amd import engine as eng
        from tests.cases import make_case

        Bl = 24
        spec, p, idx, dense, y, hp = make_case(model, B=world * Bl, D=16, **kw)
        hp = dict(hp, embedding_l2_reg=0.0, linear_l2_reg=0.0, deep_l2_reg=l2, cross_layer_l2_reg=l2, cin_l2_reg=l2)
        espec = eng.FeatureSpec(spec.sparse_names, spec.feat_sizes, spec.dense_names, linear_names=lin_names)
        dev = torch.device("cuda", 0)
        s = rd.make_sharded_engine(model, espec, 16, hp, dev, rank, world,
                                   capacity_factor=1.5 if fixed else None, micro_batches=micro)
        s.load_params({k: v for k, v in p.items() if k in s.params})
        full = torch.cat([p[f"{n}_feat_embed"] for n in spec.sparse_names])
        bias = (torch.cat([p[f"{n}_feat_bias"].reshape(-1) for n in spec.sparse_names])
                if model == "deepfm" else None)
        R = full.shape[0]
        s.st.load_global(full, bias=bias)  # linear weights start at zero (the reference's init), as on one GPU
        opt = s.optimizer("adam", 0.01)    # BEFORE any graph capture: it re-homes the dense parameters
        sl = slice(rank * Bl, (rank + 1) * Bl)
        il, dl, yl = idx[sl].cuda(), dense[sl].cuda(), y[sl].cuda()
        if segments:
            s.capture_segments(il.clone(), dl.clone(), yl.clone())
        losses = []
        for _ in range(3):
            losses.append(float(s.fwd_bwd(il, dl, yl)))
            assert not s.overflowed()
            opt.step()
        torch.save({"rows": s.st.shard[:, : 16 + 2].cpu(), "losses": losses,
                    "params": {k: v.detach().cpu() for k, v in s.params.items() if k != "table_shard"}},
                   f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model,kw,fixed,micro,segments,l2,lin_names", [
    ("deepfm", {}, True, 1, False, 0.0, None),
    ("xdeepfm", dict(cin_units=(16, 8), scale=0.2), False, 2, False, 1e-3, None),
    ("dcn", dict(cross_layers=2, scale=0.15), True, 2, True, 1e-3, ["C1", "I0", "C3"]),
    ("deepfm", {}, True, 2, True, 0.0, None),
    ("dcn", dict(cross_layers=2, scale=0.15), True, 1, True, 1e-3, None),
])
def test_two_ranks_three_optimizer_steps_equal_single_gpu(hip_lib, tmp_path, model, kw, fixed, micro, segments, l2,
                                                          lin_names):
    """The sharded TRAINING step: fwd+bwd, owner-side row-wise Adam on each rank's shard from the gradient
    rows it received, Adam on the all-reduced dense gradients - three steps on two ranks equal three steps
    of the single-GPU engine with optim.SparseTableOptimizer + FusedDenseOptimizer on the whole batch
    (same lazy-Adam semantics), to 1e-6; with the dense parameters' l2 terms (added once, not once per rank
    and micro-batch) and a linear_features subset."""
    from recman_amd import engine as eng
    from recman_amd.optim import FusedDenseOptimizer, SparseTableOptimizer
    from tests.cases import make_case

    world, Bl = 2, 24
    out = str(tmp_path / "t")
    port = _port(29900, repr((model, fixed, micro, segments, l2, lin_names)))
    mp.spawn(_train_worker, args=(world, port, model, kw, fixed, micro, segments, l2, lin_names, out),
             nprocs=world, join=True)
    res = [torch.load(f"{out}.{r}", weights_only=True) for r in range(world)]

    spec, p, idx, dense, y, hp = make_case(model, B=world * Bl, D=16, **kw)
    hp = dict(hp, embedding_l2_reg=0.0, linear_l2_reg=0.0, deep_l2_reg=l2, cross_layer_l2_reg=l2, cin_l2_reg=l2)
    e = eng.ENGINES[model](eng.FeatureSpec(spec.sparse_names, spec.feat_sizes, spec.dense_names,
                                           linear_names=lin_names), 16, hp)
    e.load_params({k: v for k, v in p.items() if k in e.params})   # (no linear_w: it starts at zero)
    sopt, dopt = SparseTableOptimizer(e, "adam", 0.01), FusedDenseOptimizer(e, "adam", 0.01)
    ib, db, yb = idx.cuda(), dense.cuda(), y.cuda()
    losses = []
    for _ in range(3):
        losses.append(float(e.fwd_bwd(ib, db, yb)))
        sopt.step(ib)
        dopt.step()
    D = 16
    for r in range(world):
        want = e.rows[r::world, : D + 2].cpu()
        diff = (res[r]["rows"] - want).abs()
        err = float(diff.max())
        assert err <= 1e-6 * max(1.0, float(want.abs().max())), (
            f"rank {r} shard rows after 3 steps: {err:.3e}; rows off: {torch.nonzero(diff.max(1).values > 1e-6).reshape(-1).tolist()[:20]}"
            f" of {diff.shape[0]}; per-column max {[round(float(x), 5) for x in diff.max(0).values]}")
        for k, v in res[r]["params"].items():
            w = e.params[k].detach().cpu()
            assert float((v - w).abs().max()) <= 1e-6 * max(1.0, float(w.abs().max())), (r, k)
    # the ranks' replicas of the dense parameters are bit-identical
    for k in res[0]["params"]:
        assert torch.equal(res[0]["params"][k], res[1]["params"][k]), k
    # loss of the global batch = mean of the ranks' data losses + the l2 value once
    mean_losses = [sum(res[r]["losses"][t] for r in range(world)) / world for t in range(3)]
    for t in range(3):
        assert abs(mean_losses[t] - losses[t]) < 2e-5, (t, mean_losses[t], losses[t])
    assert losses[2] < losses[0]


def _mv_case(world, Bl):
    """A DeepFM case with a multi-valued feature (C1) and a value feature (C3) over world * Bl examples."""
    from tests.cases import make_case

    spec, p, idx, dense, y, hp = make_case("deepfm", B=world * Bl, D=16, sizes=[7, 11, 5, 13, 3])
    hp = dict(hp, embedding_l2_reg=0.0, linear_l2_reg=0.0, deep_l2_reg=1e-3)
    g = torch.Generator().manual_seed(23)
    B = world * Bl
    n = torch.randint(0, 4, (B,), generator=g)           # 0..3 tags per example (empty lists included)
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64), n.cumsum(0)])
    ids = torch.randint(0, spec.feat_sizes[1], (int(n.sum()),), generator=g)
    vids = torch.randint(0, spec.feat_sizes[3], (B,), generator=g)
    vals = torch.randn(B, generator=g)
    return spec, p, idx, dense, y, hp, (offsets, ids), (vids, vals)


def _mv_slice(mvt, vt, sl, names):
    """The mv dict of the examples sl (device tensors): CSR re-based to the slice."""
    offsets, ids = mvt
    vids, vals = vt
    o = offsets[sl.start: sl.stop + 1]
    b = sl.stop - sl.start
    return {names[0]: ((o - o[0]).cuda(), ids[int(o[0]): int(o[-1])].cuda()),
            names[1]: (torch.arange(b + 1).cuda(), vids[sl].cuda(), vals[sl].cuda())}


def _mv_worker(rank, world, port, out_path, fixed=False, micro=1, segments=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("RECMAN_FORCE_COLLECTIVES", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys

        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from recman_amd import dist as rd
        from recman_amd import engine as eng

        Bl = 24
        spec, p, idx, dense, y, hp, mvt, vt = _mv_case(world, Bl)
        names = (spec.sparse_names[1], spec.sparse_names[3])
        espec = eng.FeatureSpec(spec.sparse_names, spec.feat_sizes, spec.dense_names, [names[0]], [names[1]])
        dev = torch.device("cuda", 0)
        s = rd.make_sharded_engine("deepfm", espec, 16, hp, dev, rank, world, capacity_factor=2.0 if fixed else None,
                                   micro_batches=micro, mv_capacity={names[0]: 3})
        s.load_params({k: v for k, v in p.items() if k in s.params})
        full = torch.cat([p[f"{n}_feat_embed"] for n in spec.sparse_names])
        bias = torch.cat([p[f"{n}_feat_bias"].reshape(-1) for n in spec.sparse_names])
        s.st.load_global(full, bias=bias)
        opt = s.optimizer("adam", 0.01)
        sl = slice(rank * Bl, (rank + 1) * Bl)
        il, dl, yl = idx[sl].cuda(), dense[sl].cuda(), y[sl].cuda()
        mv = _mv_slice(mvt, vt, sl, names)
        if segments:
            s.capture_segments(il.clone(), dl.clone(), yl.clone(), mv=mv)
        losses, logits = [], None
        for _ in range(3):
            losses.append(float(s.fwd_bwd(il, dl, yl, mv=mv)))
            assert not (fixed and s.overflowed())
            if logits is None:
                # (micro-batches: s.logit holds the LAST micro-batch's examples)
                logits = s.logit.clone().cpu()
            opt.step()
        torch.save({"rows": s.st.shard[:, : 16 + 2].cpu(), "losses": losses, "logit0": logits,
                    "params": {k: v.detach().cpu() for k, v in s.params.items() if k != "table_shard"}},
                   f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fixed,micro,segments", [(False, 1, False), (True, 1, False), (True, 2, False),
                                                  (True, 2, True)],
                         ids=["exact", "fixed", "fixed-micro2", "fixed-micro2-segments"])
def test_two_ranks_multi_valued_and_value_features_train_like_single_gpu(hip_lib, tmp_path, fixed, micro, segments):
    """Scratch-row features on the row-sharded table: the tags of a MultiValCsvFeat field and the ids of a
    SparseValueFeat field travel through the exchange - as an expanded occurrence list behind the plain fields'
    occurrences (exact split sizes), or as padded tag COLUMNS of the occurrence matrix (fixed-capacity layout:
    static shapes, so the micro-batch pipeline and the hipGraph segments work; rm_pool_rows_padded /
    rm_pack_pooled_grad_rows) - the pooled rows are built from the received tag rows, the tags' gradient rows go
    back to their owners: three training steps on two ranks equal three steps of the single-GPU engine (which
    pools from its local table and steps the tag rows through optim.SparseTableOptimizer) to 1e-6."""
    from recman_amd import engine as eng
    from recman_amd.optim import FusedDenseOptimizer, SparseTableOptimizer

    world, Bl = 2, 24
    out = str(tmp_path / "mv")
    port = 29871 + [(False, 1, False), (True, 1, False), (True, 2, False), (True, 2, True)].index((fixed, micro, segments))
    mp.spawn(_mv_worker, args=(world, port, out, fixed, micro, segments), nprocs=world, join=True)
    res = [torch.load(f"{out}.{r}", weights_only=True) for r in range(world)]
    spec, p, idx, dense, y, hp, mvt, vt = _mv_case(world, Bl)
    names = (spec.sparse_names[1], spec.sparse_names[3])
    e = eng.ENGINES["deepfm"](eng.FeatureSpec(spec.sparse_names, spec.feat_sizes, spec.dense_names, [names[0]],
                                              [names[1]]), 16, hp)
    e.load_params({k: v for k, v in p.items() if k in e.params})
    sopt, dopt = SparseTableOptimizer(e, "adam", 0.01), FusedDenseOptimizer(e, "adam", 0.01)
    ib, db, yb = idx.cuda(), dense.cuda(), y.cuda()
    mv = _mv_slice(mvt, vt, slice(0, world * Bl), names)
    losses, logit0 = [], None
    for _ in range(3):
        losses.append(float(e.fwd_bwd(ib, db, yb, mv=mv)))
        if logit0 is None:
            logit0 = e.logit.clone().cpu()
        sopt.step(ib)
        dopt.step()
    got0 = torch.cat([res[r]["logit0"] for r in range(world)])
    if micro == 1:
        assert float((got0 - logit0).abs().max()) <= 1e-6, "first-step logits"
    else:  # each rank's logit buffer holds its LAST micro-batch
        b = Bl // micro
        want0 = torch.cat([logit0[r * Bl + Bl - b: (r + 1) * Bl] for r in range(world)])
        assert float((got0 - want0).abs().max()) <= 1e-6, "first-step logits (last micro-batch of each rank)"
    for r in range(world):
        want = e.rows[r::world, : 16 + 2].cpu()
        err = float((res[r]["rows"] - want).abs().max())
        assert err <= 1e-6 * max(1.0, float(want.abs().max())), f"rank {r} shard rows after 3 steps: {err:.3e}"
        for k, v in res[r]["params"].items():
            w = e.params[k].detach().cpu()
            assert float((v - w).abs().max()) <= 1e-6 * max(1.0, float(w.abs().max())), (r, k)
    mean_losses = [sum(res[r]["losses"][t] for r in range(world)) / world for t in range(3)]
    for t in range(3):
        assert abs(mean_losses[t] - losses[t]) < 2e-5, (t, mean_losses[t], losses[t])
    assert losses[2] < losses[0]


def _fit_worker(rank, world, port, model, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("RECMAN_FORCE_COLLECTIVES", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys

        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import numpy as np
        from sklearn.metrics import log_loss

        import recman_amd.th as th
        from tests.test_gpu_models import ml_features, ml_frame

        df = ml_frame().iloc[:1000].copy()   # 1000 rows, batch 96: a ragged last batch (40 rows -> 20 + 20)
        genres = model in ("deepfm_genres", "deepfm_genres_fixed")
        if genres:                           # + the multi-valued `genres` feature (its tags travel through the exchange)
            from tests.test_gpu_models import GOLD

            df["genres"] = GOLD["raw_genres"][:1000].astype(object)
        fd = ml_features(df)
        if genres:
            fd["genres"] = th.MultiValCsvFeat(name="genres", tags=tuple(GOLD["genre_tags"].tolist()))
        hp = {"embedding_size": 16, "deep_dropout": (1, 1, 1), "cin_cross_layer_units": [16, 16],
              "cin_dropout": [1, 1, 1], "learning_rate": 0.01, "embedding_l2_reg": 0.0, "linear_l2_reg": 0.0,
              "deep_l2_reg": 1e-4, "micro_batches": 2}
        if model == "xdeepfm_ragged_fixed":
            # 1001 rows, batch 96: the last batch has 41 rows -> parts of 21 and 20 (one odd, one even: the
            # micro-batch choice and the bucket capacity must be the SAME on both ranks), fixed-capacity exchange
            df = ml_frame().iloc[:1001].copy()
            fd = ml_features(df)
            hp = dict(hp, exchange_capacity_factor=1.5)
        if model == "xdeepfm_defaults":
            # the reference's DEFAULT hyper-parameters (hparams/xDeepFM.py:19-34: embedding_l2_reg = linear_l2_reg =
            # 1e-5, dropout, D = 8) - the l2 terms on the table are applied lazily by the owner-side step
            hp = dict(th.hparams.xDeepFM().defaults(), learning_rate=0.01)
            assert hp["embedding_l2_reg"] == 1e-5 and hp["linear_l2_reg"] == 1e-5
        if model in ("xdeepfm", "xdeepfm_defaults", "xdeepfm_ragged_fixed"):
            m = th.xDeepFM(fd, hp, epoch=2, batch_size=96)
        elif genres:
            # "deepfm_genres_fixed": fixed-capacity buckets + the micro-batch pipeline - the tags as padded columns
            # of the occurrence matrix, their width = the widest genre list of the frame (set by fit())
            extra = dict(exchange_capacity_factor=2.0, micro_batches=2) if model == "deepfm_genres_fixed" else {}
            m = th.DeepFM(fd, embedding_size=16, embedding_l2_reg=0.0, linear_l2_reg=0.0, deep_dropout=(1, 1, 1),
                          learning_rate=0.01, epoch=2, batch_size=96)
            m.hparams.update(extra)  # (DeepFM's signature is the reference's: the exchange knobs are hparams)
            assert m._build().spec.multi_names == ["genres"]
            if extra:
                assert m._engine.st.capacity_factor == 2.0 and m._engine.micro_batches == 2
        else:
            m = th.DeepFM(fd, embedding_size=16, embedding_l2_reg=0.0, linear_l2_reg=0.0, deep_dropout=(1, 1, 1),
                          learning_rate=0.01, epoch=2, batch_size=96)
        yv = df["label"].values
        before = log_loss(yv, m.predict(df).astype(np.float64))
        assert m._shard == (rank, world) and m._engine.st.shard.shape[0] < m._engine.spec.rows
        m.fit(df, yv, random_seed_for_mini_batch=True)   # the shuffle seed is rank 0's on every rank
        if model == "deepfm_genres_fixed":
            assert m._engine._mv_T.get("genres", 0) >= 2 and m._engine.F_wide > m._engine.F  # (tags as columns)
        pred = m.predict(df)
        after = log_loss(yv, pred.astype(np.float64))
        path = f"{out_path}.ckpt"
        m.save(path)
        dist.barrier()
        m2 = (th.xDeepFM(fd, hp, epoch=1, batch_size=96) if model.startswith("xdeepfm") else
              th.DeepFM(fd, embedding_size=16, embedding_l2_reg=0.0, linear_l2_reg=0.0, deep_dropout=(1, 1, 1),
                        epoch=1, batch_size=96))
        m2.restore(path)
        pred2 = m2.predict(df)
        torch.save({"before": float(before), "after": float(after), "pred": torch.from_numpy(pred),
                    "pred_restored": torch.from_numpy(pred2),
                    "dense": {k: v.detach().cpu() for k, v in m._engine.params.items() if k != "table_shard"}},
                   f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model", ["deepfm", "xdeepfm", "deepfm_genres", "xdeepfm_defaults", "xdeepfm_ragged_fixed",
                                   "deepfm_genres_fixed"])
def test_model_fit_predict_save_restore_with_a_row_sharded_table_on_two_ranks(hip_lib, tmp_path, model):
    """The model classes under a two-rank torch.distributed job: the table row-sharded, fit() data parallel
    (ragged last batch, micro-batches, dense l2; "deepfm_genres": with ml-100k's multi-valued `genres`, whose tags
    travel through the exchange), predict() on every rank, save() / restore() with one shard file per rank.  Both ranks end with identical predictions and identical dense parameters, the loss
    went down, a restored model predicts the same."""
    world = 2
    out = str(tmp_path / "f")
    port = 29990 + ["deepfm", "xdeepfm", "deepfm_genres", "xdeepfm_defaults", "xdeepfm_ragged_fixed",
                    "deepfm_genres_fixed"].index(model)
    mp.spawn(_fit_worker, args=(world, port, model, out), nprocs=world, join=True)
    res = [torch.load(f"{out}.{r}", weights_only=True) for r in range(world)]
    assert res[0]["after"] < res[0]["before"] - 0.01
    assert torch.equal(res[0]["pred"], res[1]["pred"])
    for k in res[0]["dense"]:
        assert torch.equal(res[0]["dense"][k], res[1]["dense"][k]), k
    for r in range(world):
        assert float((res[r]["pred"] - res[r]["pred_restored"]).abs().max()) < 1e-6
    assert os.path.exists(f"{out}.ckpt.shard0of2.pt") and os.path.exists(f"{out}.ckpt.shard1of2.pt")


def test_padded_pooling_kernels_equal_the_csr_ones(hip_lib):
    """rm_pool_rows_padded / rm_pack_pooled_grad_rows (tags as padded columns, rows addressed through positions)
    against rm_pool_rows / a torch restatement of the tags' gradient rows; the router gives an empty occurrence
    (id -1) no slot."""
    from recman_amd import ops

    B, T, D, W, R = 37, 4, 16, 20, 50
    g = torch.Generator().manual_seed(4)
    n = torch.randint(0, T + 1, (B,), generator=g)
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64), n.cumsum(0)]).cuda()
    ids = torch.randint(0, R, (int(n.sum()),), generator=g).cuda()
    rows = torch.randn(R, W, generator=g).cuda()
    # padded columns inside a wider occurrence matrix, rows permuted (as after an exchange)
    wide = torch.full((B, 2 + T), -1, dtype=torch.int64).cuda()
    wide[:, :2] = 7
    seg = torch.repeat_interleave(torch.arange(B).cuda(), n.cuda())
    col = 2 + torch.arange(ids.numel()).cuda() - offsets[seg]
    wide[seg, col] = ids
    perm = torch.randperm(R, generator=g).cuda()
    recv = torch.empty_like(rows)
    recv[perm] = rows                       # row r of the table sits at position perm[r]
    pos = torch.where(wide >= 0, perm[wide.clamp(min=0)], torch.full_like(wide, -1))
    for vals in (None, torch.randn(ids.numel(), generator=g).cuda()):
        want = torch.empty(B, W).cuda()
        ops.pool_rows(rows, 0, D, offsets, ids, want, vals=vals)
        vw = None
        if vals is not None:
            vw = torch.zeros(B, 2 + T).cuda()
            vw[seg, col] = vals
        got = torch.empty(B, W).cuda()
        ops.pool_rows_padded(recv, D, pos[:, 2:], wide[:, 2:], got, vals=None if vw is None else vw[:, 2:])
        assert torch.equal(got, want)
        # gradient rows of the tags
        d_rows = torch.randn(B, 3, D, generator=g).cuda()
        gb, gl = torch.randn(B, generator=g).cuda(), torch.randn(B, generator=g).cuda()
        out = torch.zeros(R, W).cuda()
        ops.pack_pooled_grad_rows(d_rows[:, 1, :], gb, gl, D, pos[:, 2:], wide[:, 2:], out,
                                  vals=None if vw is None else vw[:, 2:])
        inv = n.clamp(min=1).float().rsqrt().cuda()[seg]
        we = vals if vals is not None else inv
        wb = torch.ones_like(inv) if vals is not None else inv
        wl = vals if vals is not None else (ids >= 1).float()
        ref = torch.zeros(R, W).cuda()
        p = perm[ids]
        # (a table row may occur twice among the tags: the LAST writer wins in both)
        for k in range(ids.numel()):
            ref[p[k], :D] = d_rows[seg[k], 1, :] * we[k]
            ref[p[k], D] = gb[seg[k]] * wb[k]
            ref[p[k], D + 1] = gl[seg[k]] * wl[k]
        uniq = torch.ones(R, dtype=torch.bool).cuda()
        cnt = torch.bincount(p, minlength=R)
        uniq[cnt > 1] = False                # rows written twice: either writer may be last on the GPU
        assert torch.allclose(out[uniq], ref[uniq], rtol=0, atol=1e-6)
    # the router: empty occurrences take no slot
    foff = torch.zeros(2 + T, dtype=torch.int64).cuda()
    world, cap = 2, 128
    p2 = torch.empty(B * (2 + T), dtype=torch.int64).cuda()
    send = torch.empty(world * cap, dtype=torch.int64).cuda()
    counts = torch.empty(world, dtype=torch.int64).cuda()
    over = torch.zeros(1, dtype=torch.int32).cuda()
    ws = torch.empty(ops._lib.lib().rm_shard_route_workspace(world), dtype=torch.int32).cuda()
    ops.shard_route_padded(wide, foff, world, cap, p2, send, counts, over, ws)
    from recman_amd.dist import route_torch

    pr, cr, sr, _ = route_torch(wide, foff, world, cap)
    assert torch.equal(p2, pr) and torch.equal(counts, cr) and torch.equal(send, sr) and int(over) == 0
    assert int((p2 < 0).sum()) == int((wide < 0).sum()) and int(counts.sum()) == int((wide >= 0).sum())
