"""GPU: the reference's layer callables (recman_amd/th/layers.py) composed exactly as the reference's models
compose them - xDeepFM._out (recman/tf/core/xDeepFM.py:49-104), DeepFM._init_graph (DeepFM.py:107-163),
DCN._init_graph (DCN.py:99-149) - against the CPU oracle: predictions 1e-5 (north star), the loss with
every layer's l2 term, and torch.autograd's gradient of every variable against the oracle's autograd
(per element, 2e-5 relative down to a tenth of the tensor's largest entry); and against the fused
engines' logits."""
import numpy as np
import pytest
import torch

from oracle import th_layers as T
from tests.cases import make_case

pytestmark = pytest.mark.gpu


def _setup(model, **kw):
    from recman_amd.th import DataInputs, DenseFeat, FeatureDictionary, SparseFeat

    spec, p, idx, dense, y, hp = make_case(model, **kw)
    fd = FeatureDictionary()
    for n, v in zip(spec.sparse_names, spec.feat_sizes):
        fd[n] = SparseFeat(n, v - 1)          # feat_size = v (null slot included, inputs.py:166)
    for n in spec.dense_names:
        fd[n] = DenseFeat(n)
    inp = DataInputs()                        # encoded inputs, as DataInputs.load leaves them
    inp.idx, inp.dense, inp.mv = idx.numpy(), dense.numpy(), {}
    for f, n in enumerate(spec.sparse_names):
        inp[n] = idx[:, f: f + 1].numpy()
    for j, n in enumerate(spec.dense_names):
        inp[n] = dense[:, j: j + 1].numpy()
    inp["y"] = y.numpy()
    return spec, p, idx, dense, y, hp, fd, inp


def _load(variables, p):
    """The oracle's parameters into the lazily created variables (same names, same shapes)."""
    with torch.no_grad():
        for k, v in variables.items():
            assert k in p, f"variable {k} has no counterpart in the oracle's parameters"
            v.copy_(p[k].reshape(v.shape).cuda())


def _check_grads(variables, grads_o):
    for k, v in variables.items():
        want = grads_o[k].double()
        got = v.grad.detach().cpu().double().reshape(want.shape)
        scale = float(want.abs().max())
        tol = 2e-5 * torch.clamp(want.abs(), min=0.1 * scale) + 1e-12
        bad = (got - want).abs() > tol
        assert not bool(bad.any()), (f"grad {k}: {int(bad.sum())} entries off, max err "
                                     f"{float((got - want).abs().max()):.3e} (tensor max {scale:.3e})")


def test_xdeepfm_out_composed_from_layers(hip_lib):
    from recman_amd import engine as eng
    from recman_amd.th import layers as L

    spec, p, idx, dense, y, hp, fd, inp = _setup("xdeepfm", B=70, D=8, cin_units=(16, 8), scale=0.2)
    variables = {}

    def out(training=True):  # xDeepFM._out, line for line
        emb = L.FeatEmbeddingLayer(variables, fd, 8, hp["embedding_l2_reg"], use_bias=False, seed=2019)
        feat_embeds, _ = emb(inp)
        linear_feats = fd.linear_feats
        linear_inputs = L.SparseLinearCombiner(linear_feats)(inp)
        linear = L.SparseLinearLayer(variables, linear_feats, hp["linear_l2_reg"], training=training)
        linear_logit = linear(linear_inputs)
        cin = L.CIN(variables, hp["cin_cross_layer_units"], hp["cin_activation"], [1, 1, 1], hp["cin_l2_reg"])
        cin_logit = cin(feat_embeds)
        dnn_input = L.DNNCombiner()([feat_embeds] + inp.dense_inputs(fd))
        dnn = L.DNN(variables, hp["deep_hidden_units"], [1, 1, 1], hp["deep_activation"], hp["deep_l2_reg"])
        dnn_logit = dnn(dnn_input)
        final_logit = linear_logit + cin_logit + dnn_logit
        return L.PredictionLayer(variables, "classification")(final_logit), final_logit, [emb, linear, dnn, cin]

    out()                      # first call creates the variables (reference names)
    assert {"C0_feat_embed", "linear_w", "linear_w0", "dnn_layer_0_weights", "dnn_layer_1_bias", "dnn_w", "dnn_w0",
            "cin_filter_0", "cin_bias_1", "cin_w", "cin_w0"} <= set(variables)
    assert not any(k.endswith("_feat_bias") for k in variables)   # use_bias=False (xDeepFM.py:54)
    assert float(variables["linear_w"].detach().abs().max()) == 0.0 and float(variables["cin_w"].detach().abs().max()) > 0
    _load(variables, p)
    pred, logit, layers = out()
    loss = L.create_loss(inp.y, pred) + sum(layer.l2() for layer in layers)   # xDeepFM._loss
    loss.backward()

    loss_o, logit_o, pred_o, grads_o = T.fwd_bwd("xdeepfm", p, spec, idx, dense, y, hp)
    assert float((logit.detach().cpu().reshape(-1) - logit_o).abs().max()) < 1e-5
    assert float((pred.detach().cpu() - pred_o).abs().max()) < 1e-6
    assert abs(float(loss) - float(loss_o)) < 1e-5
    _check_grads(variables, grads_o)
    # ... and the fused engine on the same weights
    e = eng.XDeepFMEngine(eng.FeatureSpec(spec.sparse_names, spec.feat_sizes, spec.dense_names), 8, hp)
    e.load_params({k: v for k, v in p.items() if k in e.params or k == "linear_w"})
    e.forward(idx.cuda(), dense.cuda(), training=True)
    assert float((e.logit - logit.detach().reshape(-1)).abs().max()) < 1e-5
    # an optimizer over `variables` is the reference's optimizer.minimize(..., variables.values())
    opt = torch.optim.SGD(variables.values(), lr=0.1)
    opt.step()
    pred2, _, _ = out()
    assert float(L.create_loss(inp.y, pred2)) < float(L.create_loss(inp.y, pred.detach()))


def test_deepfm_graph_composed_from_layers_with_fm_and_manual_weights(hip_lib):
    from recman_amd.th import layers as L

    spec, p, idx, dense, y, hp, fd, inp = _setup("deepfm", B=53, D=8)
    variables = {}

    def out(training=True):  # DeepFM._init_graph: linear + fm + dnn
        emb = L.FeatEmbeddingLayer(variables, fd, 8, hp["embedding_l2_reg"], use_bias=True)
        E, bias = emb(inp)
        feats = fd.linear_feats
        lin = L.LinearLayer(variables, feats, hp["linear_l2_reg"], training=training)
        linear_logit = lin(L.LinearCombiner(feats)(inp))
        fm_logit = L.FMLayer(dropout=(1.0, 1.0))(E, bias)
        dnn = L.DNN(variables, hp["deep_hidden_units"], (1, 1, 1), hp["deep_activation"], hp["deep_l2_reg"])
        dnn_logit = dnn(L.DNNCombiner()([E] + inp.dense_inputs(fd)))
        final = linear_logit + fm_logit + dnn_logit
        return L.PredictionLayer(variables, "classification")(final), final, [emb, lin, dnn]

    out()
    _load(variables, p)
    pred, logit, layers = out()
    loss = L.create_loss(inp.y, pred) + sum(layer.l2() for layer in layers)
    loss.backward()
    loss_o, logit_o, pred_o, grads_o = T.fwd_bwd("deepfm", p, spec, idx, dense, y, hp)
    assert float((logit.detach().cpu().reshape(-1) - logit_o).abs().max()) < 1e-5
    assert abs(float(loss) - float(loss_o)) < 1e-5
    _check_grads(variables, grads_o)
    # training=False adds the features' manual weights to linear_w (layers.py:338-345)
    fd[spec.sparse_names[0]].encoder = None
    fd[spec.sparse_names[0]].set_weights({2: -3.0})
    pred_i, logit_i, _ = out(training=False)
    shift = (logit_i - logit).detach().cpu().reshape(-1)
    want = torch.where(idx[:, 0] == 2, torch.tensor(-3.0), torch.tensor(0.0))
    assert float((shift - want).abs().max()) < 1e-5


def test_fm_layer_dropout_masks_scale_and_gradcheck(hip_lib):
    from recman_amd.th import layers as L

    torch.manual_seed(0)
    E = torch.randn(40, 5, 8, device="cuda", requires_grad=True)
    bias = torch.randn(40, 5, 1, device="cuda", requires_grad=True)
    y = L.FMLayer()(E, bias)
    want = T.fm_layer(E.detach().cpu().double(), bias.detach().cpu().double())
    assert float((y.detach().cpu().double() - want).abs().max()) < 1e-4
    y.sum().backward()
    Ec, bc = E.detach().cpu().double().requires_grad_(True), bias.detach().cpu().double().requires_grad_(True)
    T.fm_layer(Ec, bc).sum().backward()
    assert float((E.grad.cpu().double() - Ec.grad).abs().max()) < 1e-4
    assert float((bias.grad.cpu().double() - bc.grad).abs().max()) < 1e-6
    # keep-probabilities < 1: some lookups dropped, the rest scaled by 1 / keep - the expectation stays
    out = torch.stack([L.FMLayer(dropout=(0.5, 1.0))(torch.zeros_like(E), torch.ones_like(bias)) for _ in range(200)])
    assert abs(float(out.mean()) - 5.0) < 0.2 and float(out.std()) > 0.5


def test_dcn_graph_composed_from_layers_with_crossnet(hip_lib):
    from recman_amd.th import layers as L

    spec, p, idx, dense, y, hp, fd, inp = _setup("dcn", B=61, D=8, cross_layers=3, scale=0.15, hidden=(40, 24))
    variables = {}

    def out():  # DCN._init_graph: dnn + cross (+ linear)
        emb = L.FeatEmbeddingLayer(variables, fd, 8, hp["embedding_l2_reg"], use_bias=False)
        E, _ = emb(inp)
        dnn_input = L.DNNCombiner()([E] + inp.dense_inputs(fd))
        dnn = L.DNN(variables, hp["deep_hidden_units"], (1, 1, 1), hp["deep_activation"], hp["deep_l2_reg"])
        dnn_logit = dnn(dnn_input)
        cn = L.CrossNet(variables, hp["cross_layer_num"], hp["cross_layer_l2_reg"])
        cn_logit = cn(dnn_input)
        feats = fd.linear_feats
        lin = L.LinearLayer(variables, feats, hp["linear_l2_reg"])
        final = dnn_logit + cn_logit + lin(L.LinearCombiner(feats)(inp))
        assert len(cn.weights) == 3
        return L.PredictionLayer(variables, "classification")(final), final, [emb, dnn, cn, lin]

    out()
    assert variables["cross_w"].shape == (3, 8 * 5 + 2) and variables["cross_w_out"].shape == (42, 1)
    _load(variables, p)
    pred, logit, layers = out()
    loss = L.create_loss(inp.y, pred) + sum(layer.l2() for layer in layers)
    loss.backward()
    loss_o, logit_o, pred_o, grads_o = T.fwd_bwd("dcn", p, spec, idx, dense, y, hp)
    assert float((logit.detach().cpu().reshape(-1) - logit_o).abs().max()) < 1e-5
    assert abs(float(loss) - float(loss_o)) < 1e-5
    _check_grads(variables, grads_o)


def test_layers_reject_what_they_do_not_cover(hip_lib):
    from recman_amd.th import FeatureDictionary, SparseFeat
    from recman_amd.th import layers as L

    with pytest.raises(AssertionError):
        L.CIN({}, [8, 8], "relu", [1, 0.9])           # one keep-probability per layer + the input (layers.py:657)

    class SequenceFeat:                                # (raises in the reference too, inputs.py:443)
        name, feat_size = "q", 4

    with pytest.raises(NotImplementedError):
        L.FeatEmbedding({}, SequenceFeat(), 8)
    fd2 = FeatureDictionary()
    fd2["s"] = SparseFeat("s", 3)
    with pytest.raises(ValueError):
        L.FeatEmbeddingLayer({}, fd2, 6)
    with pytest.raises(ValueError):
        L.create_loss(np.zeros(3), torch.zeros(3), task="ranking")


def test_xdeepfm_from_layers_with_cin_dropout_and_a_multi_valued_feature(hip_lib, monkeypatch):
    """th.layers.CIN with dropout (layers.py:707-708, 740) and FeatEmbeddingLayer / SparseLinearCombiner with a
    MultiValCsvFeat (sqrtn-pooled lookup layers.py:144-169, multi-hot linear input utils.py:86-108) - composed as
    xDeepFM._out composes them, against the oracle run with the SAME dropout masks."""
    from recman_amd.th import MultiValCsvFeat
    from recman_amd.th.inputs import CSR
    from recman_amd.th import layers as L

    spec, p, idx, dense, y, hp, fd, inp = _setup("xdeepfm", B=45, D=8, cin_units=(16, 8), scale=0.2)
    mname = spec.sparse_names[2]
    V = spec.feat_sizes[2]
    ospec = T.Spec(spec.sparse_names, spec.feat_sizes, spec.dense_names, multi_names=[mname])
    g = torch.Generator().manual_seed(3)
    n = torch.randint(0, 4, (45,), generator=g)
    n[1] = 0
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64), n.cumsum(0)])
    ids = torch.randint(0, V, (int(n.sum()),), generator=g)
    mv = {mname: (offsets, ids)}
    # the feature dictionary with the multi-valued feature in the same position
    fd2 = type(fd)()
    for k, f in fd.items():
        fd2[k] = MultiValCsvFeat(name=mname, tags=tuple(f"t{i}" for i in range(V - 1))) if k == mname else f
    assert fd2[mname].feat_size == V
    inp.mv = {mname: CSR(offsets.numpy(), ids.numpy())}
    inp.idx = inp.idx.copy()
    inp.idx[:, 2] = 0
    # deterministic dropout masks, shared with the oracle: input E, layer 0 maps, layer 1 maps
    keep = [0.8, 0.7, 1.0]
    shapes = [(45, 5, 8), (45, 16, 8), (45, 8, 8)]
    gm = torch.Generator().manual_seed(4)
    masks = [(torch.rand(*sh, generator=gm) < k).float() if k < 1 else None for sh, k in zip(shapes, keep)]
    drawn = iter([m for m in masks if m is not None])
    monkeypatch.setattr(L, "_keep_mask", lambda shape, k: next(drawn).cuda())
    variables = {}

    def out():
        emb = L.FeatEmbeddingLayer(variables, fd2, 8, hp["embedding_l2_reg"], use_bias=False, seed=2019)
        feat_embeds, _ = emb(inp)
        linear_feats = fd2.linear_feats
        linear = L.SparseLinearLayer(variables, linear_feats, hp["linear_l2_reg"])
        linear_logit = linear(L.SparseLinearCombiner(linear_feats)(inp))
        cin = L.CIN(variables, hp["cin_cross_layer_units"], hp["cin_activation"], keep, hp["cin_l2_reg"])
        cin_logit = cin(feat_embeds)
        dnn = L.DNN(variables, hp["deep_hidden_units"], [1, 1, 1], hp["deep_activation"], hp["deep_l2_reg"])
        dnn_logit = dnn(L.DNNCombiner()([feat_embeds] + inp.dense_inputs(fd2)))
        final_logit = linear_logit + cin_logit + dnn_logit
        return L.PredictionLayer(variables, "classification")(final_logit), final_logit, [emb, linear, dnn, cin]

    monkeypatch.setattr(L, "_keep_mask", lambda shape, k: torch.ones(*shape, device="cuda"))
    out()  # creates the variables
    # the oracle's parameters; linear_w random in the layers' own order (the reference's: sparse, value,
    # multi-valued, dense features, utils.py:31-36 - the oracle spec stacks it the same way and reads it back
    # from the variables below)
    with torch.no_grad():
        for k, v in variables.items():
            if k == "linear_w":
                v.copy_(0.2 * torch.randn(v.shape, generator=torch.Generator().manual_seed(8)).cuda())
            else:
                v.copy_(p[k].reshape(v.shape).cuda())
    drawn = iter([m for m in masks if m is not None])
    monkeypatch.setattr(L, "_keep_mask", lambda shape, k: next(drawn).cuda())
    pred, logit, layers = out()
    loss = L.create_loss(inp.y, pred) + sum(layer.l2() for layer in layers)
    loss.backward()
    hp_o = dict(hp, cin_dropout=keep)
    p_o = {k: v.detach().cpu() for k, v in variables.items()}
    loss_o, logit_o, pred_o, grads_o = T.fwd_bwd("xdeepfm", p_o, ospec, idx, dense, y, hp_o,
                                                 masks={"cin": masks}, mv=mv)
    assert float((logit.detach().cpu().reshape(-1) - logit_o).abs().max()) < 1e-5
    assert abs(float(loss) - float(loss_o)) < 1e-5
    _check_grads(variables, grads_o)
