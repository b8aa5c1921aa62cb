"""CPU: pins the oracle (KATs, numpy-f64 twin vs torch, closed forms, gradcheck)."""
import numpy as np
import pytest
import torch

from oracle import np_layers as N
from oracle import th_layers as T
from tests.cases import make_case


def test_cin_notebook_kat():
    # recman/notes/xDeepFM.ipynb cell 6: X0 = [[1,2,3,4],[5,6,7,8]], all-ones filters,
    # units (16,16), no activation/bias.  The notebook stores no outputs; expected
    # values are hand-derived.
    E = np.array([[[1, 2, 3, 4], [5, 6, 7, 8]]], dtype=np.float64)
    f0, f1 = np.ones((1, 4, 16)), np.ones((1, 16, 16))
    _, pooled, maps = N.cin(E, [f0, f1], [np.zeros(16), np.zeros(16)], np.ones((24, 1)),
                            np.zeros(1), activation=None, return_maps=True)
    assert np.array_equal(maps[0][0, 0], [36, 64, 100, 144])
    assert np.array_equal(maps[1][0, 0], [1728, 4096, 8000, 13824])
    assert np.array_equal(pooled[0], [344] * 8 + [27648] * 16)
    # same through the torch oracle
    p = {"cin_filter_0": torch.ones(1, 4, 16), "cin_bias_0": torch.zeros(16),
         "cin_filter_1": torch.ones(1, 16, 16), "cin_bias_1": torch.zeros(16),
         "cin_w": torch.ones(24, 1), "cin_w0": torch.zeros(1)}
    _, pooled_t, _ = T.cin(p, torch.tensor(E, dtype=torch.float32), 2, activation=None,
                           return_maps=True)
    assert np.array_equal(pooled_t.numpy()[0], [344] * 8 + [27648] * 16)


def test_cin_z_layout_is_i_major():
    # layers.py:721-726: Z[b,d,i*H+j] = X0[b,i,d]*Xk[b,j,d]  ->  Z[0,0,:] = [1,5,5,25]
    x0 = np.array([[[1, 2, 3, 4], [5, 6, 7, 8]]], dtype=np.float64)
    W = np.zeros((1, 4, 4))
    W[0] = np.eye(4)  # identity filter exposes Z
    _, _, maps = N.cin(x0, [W], [np.zeros(4)], np.ones((4, 1)), np.zeros(1), activation=None,
                       return_maps=True)
    assert np.array_equal(maps[0][0, :, 0], [1, 5, 5, 25])


def test_fm_pairwise_identity_exact():
    # y2 = sum_{f<g} e_f . e_g, exact on small integers in fp32
    rng = np.random.default_rng(0)
    E = rng.integers(-4, 5, size=(6, 5, 8)).astype(np.float32)
    bias = rng.integers(-3, 4, size=(6, 5, 1)).astype(np.float32)
    want = np.zeros((6, 1), np.float32)
    for f in range(5):
        for g in range(f + 1, 5):
            want[:, 0] += (E[:, f] * E[:, g]).sum(-1)
    want += bias.sum(1)
    assert np.array_equal(N.fm_layer(E, bias), want)
    assert np.array_equal(T.fm_layer(torch.tensor(E), torch.tensor(bias)).numpy(), want)


def test_cross_closed_form_integers():
    x0 = np.array([[1., 2., -1.], [0., 3., 2.]])
    ws = np.array([[1., 0., 2.], [-1., 1., 0.]])
    bs = np.array([[1., 1., 0.], [0., -2., 1.]])
    wo = np.array([[1.], [2.], [-1.]])
    # by hand: example 0: s0 = 1-2 = -1 -> x1 = -x0 + b0 + x0 = b0 = [1,1,0]; s1 = -1+1 = 0 -> x2 = b1 + x1 = [1,-1,1]
    #          logit = 1 - 2 - 1 = -2
    out = N.cross_net(x0, ws, bs, wo)
    assert out[0, 0] == -2.0
    p = {"cross_w": torch.tensor(ws), "cross_b": torch.tensor(bs), "cross_w_out": torch.tensor(wo)}
    assert np.allclose(T.cross_net(p, torch.tensor(x0)).numpy(), out)


@pytest.mark.parametrize("model,kw", [("deepfm", {}), ("dcn", dict(cross_layers=3)),
                                       ("xdeepfm", dict(cin_units=(12, 10)))])
def test_numpy_f64_twin_matches_torch(model, kw):
    spec, p, idx, dense, y, hp = make_case(model, dtype=torch.float64, **kw)
    pn = {k: v.numpy() for k, v in p.items()}
    tabs = [pn[f"{n}_feat_embed"] for n in spec.sparse_names]
    E, _ = N.feat_embedding_layer(idx.numpy(), tabs)
    x = N.dnn_combiner(E, dense.numpy())
    oh = N.linear_one_hot_input(idx.numpy(), spec.feat_sizes, dense.numpy(), np.float64)
    lin = N.linear_layer(oh, pn["linear_w"], pn["linear_w0"])
    nl = len(hp["deep_hidden_units"])
    dn = N.dnn(x, [pn[f"dnn_layer_{i}_weights"] for i in range(nl)],
               [pn[f"dnn_layer_{i}_bias"] for i in range(nl)], pn["dnn_w"], pn["dnn_w0"],
               hp["deep_activation"])
    if model == "deepfm":
        bt = [pn[f"{n}_feat_bias"] for n in spec.sparse_names]
        _, bias = N.feat_embedding_layer(idx.numpy(), tabs, bt)
        logit = lin + N.fm_layer(E, bias) + dn
    elif model == "dcn":
        logit = dn + N.cross_net(x, pn["cross_w"], pn["cross_b"], pn["cross_w_out"]) + lin
    else:
        nc = len(hp["cin_cross_layer_units"])
        logit = lin + dn + N.cin(E, [pn[f"cin_filter_{i}"] for i in range(nc)],
                                 [pn[f"cin_bias_{i}"] for i in range(nc)], pn["cin_w"], pn["cin_w0"])
    want = T.MODELS[model][0](p, spec, idx, dense, hp).numpy()
    assert np.allclose(logit, want, rtol=1e-12, atol=1e-12)
    pred = N.prediction(logit)
    loss = N.binary_crossentropy(y.numpy(), pred)
    loss_t = T.create_loss(y, T.prediction(torch.tensor(logit)))
    assert abs(loss - float(loss_t)) < 1e-12


def test_linear_manual_weights_predict_path():
    # layers.py:338-345: training=False adds per-feature manual weights to W
    spec, p, idx, dense, y, hp = make_case("deepfm", dtype=torch.float64)
    mw = torch.zeros(p["linear_w"].shape[0], dtype=torch.float64)
    mw[3] = -5.0
    a = T.linear_layer(p, spec, idx, dense)
    b = T.linear_layer(p, spec, idx, dense, manual_weights=mw)
    hit = (idx[:, 0] == 3).double().reshape(-1, 1) * -5.0
    assert torch.allclose(b - a, hit)


@pytest.mark.parametrize("model,kw", [("deepfm", {}), ("dcn", dict(cross_layers=2)),
                                       ("xdeepfm", dict(cin_units=(6, 4)))])
def test_oracle_gradients_finite_difference(model, kw):
    spec, p, idx, dense, y, hp = make_case(model, B=9, F=3, D=4, Dn=1, hidden=(5,), dtype=torch.float64,
                                           **kw)
    loss, _, _, grads = T.fwd_bwd(model, p, spec, idx, dense, y, hp)
    rng = np.random.default_rng(1)
    for name in list(p)[:: max(1, len(p) // 6)]:
        flat = p[name].reshape(-1)
        k = int(rng.integers(0, flat.numel()))
        eps = 1e-6
        old = float(flat[k])
        flat[k] = old + eps
        lp, _, _ = T.model_loss(model, p, spec, idx, dense, y, hp)
        flat[k] = old - eps
        lm, _, _ = T.model_loss(model, p, spec, idx, dense, y, hp)
        flat[k] = old
        fd = (float(lp) - float(lm)) / (2 * eps)
        assert abs(fd - float(grads[name].reshape(-1)[k])) < 1e-6, name


def test_keras_bce_clip_semantics():
    p = np.array([0.0, 1.0, 0.5, 1e-9], dtype=np.float64)
    y = np.array([1, 0, 1, 0])
    got = N.binary_crossentropy(y, p)
    e = 1e-7
    want = np.mean([-np.log(e + e), -np.log(1 - (1 - e) + e), -np.log(0.5 + e), -np.log(1 - e + e)])
    assert abs(got - want) < 1e-12


def test_multi_val_sqrtn_pooling_and_multi_hot_linear():
    # layers.py:150-156 (combiner="sqrtn") and utils.py:86-108 (multi-hot, slot 0 zeroed)
    spec = T.Spec(["a", "g", "b"], [5, 4, 6], ["d0"], multi_names=["g"])
    assert spec.lin_offsets == ([0, 11, 5], 15)
    p = T.make_params(spec, "deepfm", 4, hidden=(4,), scale=0.3, dtype=torch.float64)
    idx = torch.tensor([[1, 0, 2], [3, 0, 5], [0, 0, 0]])
    dense = torch.randn(3, 1, dtype=torch.float64)
    mv = {"g": (torch.tensor([0, 2, 2, 5]), torch.tensor([1, 3, 0, 2, 2]))}
    E, bias = T.feat_embedding_layer(p, spec, idx, True, mv)
    t, bt = p["g_feat_embed"], p["g_feat_bias"]
    assert torch.allclose(E[0, 1], (t[1] + t[3]) / 2 ** 0.5)
    assert float(E[1, 1].abs().sum()) == 0.0                      # no tags -> zeros
    assert torch.allclose(E[2, 1], (t[0] + 2 * t[2]) / 3 ** 0.5)  # unknown tag looks up row 0
    assert torch.allclose(bias[2, 1], (bt[0] + 2 * bt[2]) / 3 ** 0.5)
    lin = T.linear_layer(p, spec, idx, dense, mv=mv)
    W = p["linear_w"]
    want2 = p["linear_w0"] + W[0 + 0] + W[5 + 0] + 2 * W[11 + 2] + dense[2] @ W[15:16]  # id 0 contributes nothing
    assert torch.allclose(lin[2], want2.reshape(-1))


def test_value_feature_reduces_to_plain_lookup_and_scales():
    """SparseValueFeat branch of the oracle (layers.py:129-142, utils.py:70-71): with value 1 it is
    the plain SparseFeat lookup; the embedding and linear terms scale with the value, the FM
    bias does not."""
    spec, p, idx, dense, y, hp = make_case("deepfm", B=9, D=4)
    name = spec.sparse_names[1]
    vspec = T.Spec(spec.sparse_names, spec.feat_sizes, spec.dense_names, value_names=[name])
    # the linear_w block order changes (sparse, value, ...): permute the plain weights accordingly
    offs_plain, _ = spec.lin_offsets
    offs_val, _ = vspec.lin_offsets
    pv = dict(p)
    W = p["linear_w"].clone()
    for f, n in enumerate(spec.sparse_names):
        W[offs_val[f]: offs_val[f] + spec.feat_sizes[f]] = p["linear_w"][offs_plain[f]: offs_plain[f] + spec.feat_sizes[f]]
    pv["linear_w"] = W
    ones = torch.ones(9)
    plain = T.deepfm_logit(p, spec, idx, dense, hp, training=False)
    as_val = T.deepfm_logit(pv, vspec, idx, dense, hp, training=False, mv={name: (idx[:, 1], ones)})
    assert torch.allclose(plain, as_val, atol=1e-6)
    E1, b1 = T.feat_embedding_layer(pv, vspec, idx, True, mv={name: (idx[:, 1], ones)})
    E3, b3 = T.feat_embedding_layer(pv, vspec, idx, True, mv={name: (idx[:, 1], 3 * ones)})
    assert torch.allclose(E3[:, 1], 3 * E1[:, 1]) and torch.equal(b3, b1)
    assert torch.equal(E3[:, 0], E1[:, 0])


def test_linear_features_subset_layout_and_value():
    """get_linear_features with a name list (utils.py:27-30): linear_w stacks only those features'
    one-hot blocks / columns, in the order given."""
    spec = T.Spec(["a", "b", "c"], [4, 3, 5], ["x", "y"], linear_names=["y", "c", "a"])
    offs, doffs, total = spec.lin_layout
    assert (offs, doffs, total) == ([6, None, 1], [None, 0], 10)
    W = torch.arange(10, dtype=torch.float64).reshape(10, 1) + 1.0
    p = {"linear_w": W, "linear_w0": torch.tensor([0.5], dtype=torch.float64)}
    idx = torch.tensor([[2, 1, 4], [0, 2, 0]])
    dense = torch.tensor([[10.0, 2.0], [20.0, -1.0]], dtype=torch.float64)
    out = T.linear_layer(p, spec, idx, dense)
    # y*W[0] + W[1 + idx_c] + W[6 + idx_a] + w0
    want = torch.tensor([[2.0 * 1 + (1 + 1 + 4) + (6 + 1 + 2) + 0.5], [-1.0 * 1 + 2 + 7 + 0.5]],
                        dtype=torch.float64)
    assert torch.equal(out, want)
    with pytest.raises(ValueError):
        T.Spec(["a"], [4], [], linear_names=["zz"]).lin_layout
