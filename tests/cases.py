"""Seeded synthetic cases shared by the CPU and GPU tests (test infrastructure)."""
import torch

from oracle import th_layers as T


def make_case(model, B=37, F=5, D=8, Dn=2, sizes=None, hidden=(16, 8), cin_units=(), cross_layers=0,
              seed=0, scale=0.3, hp_extra=None, dtype=torch.float32):
    if sizes is None:
        sizes = [7, 11, 5, 13, 3, 17, 4, 9, 6, 8][:F] if F <= 10 else [5 + (i * 7) % 23 for i in range(F)]
    spec = T.Spec([f"C{i}" for i in range(F)], sizes, [f"I{j}" for j in range(Dn)])
    p = T.make_params(spec, model, D, hidden=hidden, cin_units=cin_units, cross_layers=cross_layers,
                      use_bias=(model == "deepfm"), seed=2019 + seed, dtype=dtype, scale=scale)
    g = torch.Generator().manual_seed(seed)
    idx = torch.stack([torch.randint(0, v, (B,), generator=g) for v in spec.feat_sizes], 1)
    dense = torch.randn(B, Dn, generator=g, dtype=dtype)
    y = (torch.rand(B, generator=g) < 0.3).long()
    hp = dict(deep_hidden_units=tuple(hidden), embedding_l2_reg=1e-3, linear_l2_reg=1e-3,
              deep_l2_reg=1e-3)
    if model == "deepfm":
        hp.update(deep_activation="relu")
    if model == "dcn":
        hp.update(deep_activation="relu", cross_layer_num=cross_layers, cross_layer_l2_reg=1e-3)
    if model == "xdeepfm":
        hp.update(deep_activation="leaky_relu", cin_activation="leaky_relu",
                  cin_cross_layer_units=tuple(cin_units), cin_l2_reg=1e-3)
    hp.update(hp_extra or {})
    return spec, p, idx, dense, y, hp
