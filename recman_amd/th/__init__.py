"""The `recman.th`-shaped surface: the PyTorch backend the reference leaves as an empty
stub (recman/th/layers.py is 0 bytes, recman/th/DeepFM.py:12-13 is `pass`), here backed
by hand-written gfx950 kernels.  Same class names, constructor arguments and
fit()/predict()/evaluate() signatures as recman/tf/core."""
from .BestModelFinder import BestModelFinder
from .DCN import DCN
from .DeepFM import DeepFM
from .DeepModel import DeepModel
from .inputs import (DataInputs, DenseFeat, FeatureDictionary, MultiValCsvFeat, ResilientLabelEncoder,
                     SparseFeat, SparseValueFeat)
from .xDeepFM import xDeepFM
from . import hparams
from . import layers

__all__ = ["BestModelFinder", "DCN", "DeepFM", "DeepModel", "xDeepFM", "DataInputs", "DenseFeat", "FeatureDictionary",
           "MultiValCsvFeat", "ResilientLabelEncoder", "SparseFeat", "SparseValueFeat", "hparams", "layers"]
