"""Hyper-parameter dictionaries with the reference's keys and defaults
(recman/tf/hparams/xDeepFM.py:7-34, BaseHyperParameters.py:67-100), minus TensorBoard:
`HParam(domain)` keeps a plain list of candidate values and grid_search() yields dicts."""
import itertools


class HParam:
    def __init__(self, name, default_value):
        assert name
        self.name = name
        self.default_value = default_value
        self.hp_domain = [default_value]

    def __call__(self, domain=None):
        self.hp_domain = list(domain) if domain is not None else [self.default_value]
        return self


class BaseHyperParameters(dict):
    LearningRate = "learning_rate"
    Optimizer = "optimizer"

    def __init__(self):
        dict.__init__(self)
        self.add_param(self.LearningRate, 0.001)
        self.add_param(self.Optimizer, "adam")

    def add_param(self, name, default_val):
        self[name] = HParam(name, default_val)()

    def grid_search(self, print_hp=False):
        names = list(self.keys())
        for combo in itertools.product(*[self[n].hp_domain for n in names]):
            bag = dict(zip(names, combo))
            if print_hp:
                print(bag)
            yield bag

    def defaults(self):
        return {k: v.default_value for k, v in self.items()}


class xDeepFM(BaseHyperParameters):
    EmbeddingSize = "embedding_size"
    EmbeddingL2Reg = "embedding_l2_reg"
    LinearL2Reg = "linear_l2_reg"
    LinearFeatures = "linear_features"
    DeepHiddenUnits = "deep_hidden_units"
    DeepDropOut = "deep_dropout"
    DeepActivation = "deep_activation"
    DeepL2Reg = "deep_l2_reg"
    CinCrossLayerUnits = "cin_cross_layer_units"
    CinDropOut = "cin_dropout"
    CinActivation = "cin_activation"
    CinL2Reg = "cin_l2_reg"

    def __init__(self):
        BaseHyperParameters.__init__(self)
        self.add_param(self.EmbeddingSize, 8)
        self.add_param(self.EmbeddingL2Reg, 0.00001)
        self.add_param(self.LinearL2Reg, 0.00001)
        self.add_param(self.LinearFeatures, [])
        self.add_param(self.DeepHiddenUnits, (32, 32))
        self.add_param(self.DeepDropOut, (0.8, 0.8, 0.8))
        self.add_param(self.DeepActivation, "leaky_relu")
        self.add_param(self.DeepL2Reg, 0.00001)
        self.add_param(self.CinCrossLayerUnits, [100, 100, 100])
        self.add_param(self.CinDropOut, [1, 1, 1, 1])
        self.add_param(self.CinActivation, "leaky_relu")
        self.add_param(self.CinL2Reg, 0.00001)
