"""GPU: variable initialisation.  engine.init_reference follows the reference's
distributions - TF's RNG stream itself cannot be reproduced, so the parity tests inject weights; what CAN
be pinned is the distribution of every variable class:
  * embedding tables, DNN / CIN weights: glorot_normal = truncated normal(0, sqrt(2 / (fan_in + fan_out))),
    resampled outside +-2 sigma (recman/tf/core/utils.py:156-183; layers.py:95-110, 532-574, 659-695);
    a +-2-sigma truncated normal has std 0.87962566 * sigma;
  * cin_w: glorot_uniform, bounds +-sqrt(6 / (fan_in + fan_out)) (utils.py:186-189, layers.py:690);
  * everything else (bias tables, biases, linear_w / linear_w0, dnn_w0, cin_w0): zeros
    (layers.py:106-110, 318-328, 541-574, 673-695)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

TRUNC_STD = 0.87962566103423978  # std of a standard normal truncated to [-2, 2]


def _check_trunc_normal(t, fan_in, fan_out, what, rel=0.02):
    sigma = math.sqrt(2.0 / (fan_in + fan_out))
    assert float(t.abs().max()) <= 2 * sigma * (1 + 1e-6), f"{what}: a value beyond 2 sigma"
    if t.numel() >= 20000:
        std = float(t.double().std())
        assert abs(std / (TRUNC_STD * sigma) - 1) < rel, f"{what}: std {std:.4e} vs {TRUNC_STD * sigma:.4e}"
        assert abs(float(t.double().mean())) < 0.02 * sigma, f"{what}: mean"
        assert float(t.abs().max()) > 1.9 * sigma, f"{what}: never comes near the truncation point"


@pytest.mark.parametrize("model", ["deepfm", "dcn", "xdeepfm"])
def test_init_reference_distributions(hip_lib, model):
    from recman_amd import engine as eng

    sizes, D, Dn = [30000, 5000, 7, 12000], 16, 3
    hp = dict(deep_hidden_units=(256, 128), deep_activation="relu", cross_layer_num=3,
              cin_cross_layer_units=(64, 32), cin_activation="leaky_relu")
    spec = eng.FeatureSpec([f"C{i}" for i in range(len(sizes))], sizes, [f"I{j}" for j in range(Dn)])
    e = eng.ENGINES[model](spec, D, hp)
    for base in e.storage():
        base.fill_(7.0)  # every variable must be overwritten
    eng.init_reference(e, seed=2019)
    sd = e.state_dict()
    zero_names = {"linear_w", "linear_w0", "dnn_w0", "cin_w0", "cross_b"}
    seen_trunc = 0
    for name, t in sd.items():
        if name.endswith("_feat_embed"):
            _check_trunc_normal(t, t.shape[0], t.shape[1], name)  # calc_fan of a [V, D] matrix
            seen_trunc += 1
        elif name.endswith("_weights") or name == "dnn_w":
            _check_trunc_normal(t, t.shape[0], t.shape[1], name)
            seen_trunc += 1
        elif name.startswith("cin_filter_"):
            _check_trunc_normal(t, t.shape[1], t.shape[2], name)  # [1, m*H, N]: fan of the conv kernel
            seen_trunc += 1
        elif name == "cin_w":
            bound = math.sqrt(6.0 / (t.shape[0] + t.shape[1]))
            assert float(t.abs().max()) <= bound * (1 + 1e-6) and float(t.abs().max()) > 0.5 * bound
        elif name in ("cross_w", "cross_w_out"):  # absent from the reference: glorot-normal here
            assert float(t.abs().max()) > 0
        elif name.endswith("_feat_bias") or name.endswith("_bias") or name.startswith("cin_bias_") or name in zero_names:
            assert float(t.abs().max()) == 0.0, f"{name} must start at zero"
        else:
            raise AssertionError(f"unclassified variable {name}")
    assert seen_trunc >= len(sizes) + 3
    # a second engine with another seed differs, the same seed repeats
    e2 = eng.ENGINES[model](spec, D, hp)
    eng.init_reference(e2, seed=2019)
    assert torch.equal(e2.state_dict()["C0_feat_embed"], sd["C0_feat_embed"])
    eng.init_reference(e2, seed=7)
    assert not torch.equal(e2.state_dict()["C0_feat_embed"], sd["C0_feat_embed"])
