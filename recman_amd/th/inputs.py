"""Feature descriptors and encoders: raw DataFrame columns -> int64 indices / float32
values, the producers of the `idx [B,F]` / `dense [B,Dn]` arrays the kernels consume.

Mirrors recman/tf/inputs.py (FeatureDictionary :8-43, DataInputs :46-90,
ResilientLabelEncoder :116-145, SparseFeat :148-210, DenseFeat :281-322) without any
TensorFlow type.  Re-designed for throughput: the reference re-encodes every mini-batch
with pandas `isin` + `apply(LabelEncoder.transform)`; here a column is encoded in one
vectorised pass (sorted-class search) and a whole dataset is encoded ONCE per fit().
Indices are bit-identical to the reference's for string columns (tests/test_inputs.py
checks against oracle/inputs_ref.py).
"""
from collections import OrderedDict

import numpy as np
import pandas as pd
from sklearn.preprocessing import StandardScaler

NULL_VAL = "-----"


class ResilientLabelEncoder:
    """inputs.py:116-145.  classes_ = [null_val] + sorted(unique(X)); unseen -> 0.

    Reference quirk: for an INTEGER column the reference's classes_
    becomes a string array and every id encodes to 0.  strict_reference=True
    reproduces that; the default encodes integers by value (useful behaviour)."""

    def __init__(self, null_val=NULL_VAL, strict_reference=False):
        self.null_val = null_val
        self.strict_reference = strict_reference
        self.classes_ = None
        self._sorted = None
        self._numeric = False

    def fit(self, X, y=None):
        vals = np.asarray(X)
        if vals.dtype.kind in "OUS":
            uniq = np.array(sorted(set(vals.tolist())), dtype=object)
            self._numeric = False
        else:
            uniq = np.unique(vals)
            self._numeric = True
        self._sorted = uniq
        self.classes_ = np.concatenate((np.array([self.null_val], dtype=object), uniq.astype(object)))
        return self

    def transform(self, X):
        if self._sorted is None:
            raise RuntimeError("ResilientLabelEncoder.transform before fit")
        vals = np.asarray(X)
        if self._numeric and self.strict_reference:
            return np.zeros((len(vals), 1), dtype=np.int64)
        if self._numeric != (vals.dtype.kind not in "OUS"):
            # type mismatch between fit and transform: nothing can match (as `isin` in the reference)
            return np.zeros((len(vals), 1), dtype=np.int64)
        if self._numeric:
            pos = np.searchsorted(self._sorted, vals)
            pos_c = np.minimum(pos, len(self._sorted) - 1) if len(self._sorted) else pos
            hit = (pos < len(self._sorted)) & (self._sorted[pos_c] == vals) if len(self._sorted) else np.zeros(len(vals), bool)
            return np.where(hit, pos + 1, 0).astype(np.int64).reshape(-1, 1)
        codes = pd.Categorical(vals, categories=self._sorted).codes.astype(np.int64)
        return (codes + 1).reshape(-1, 1)  # code -1 (unseen) -> 0, the null class

    def fit_transform(self, X, y=None):
        return self.fit(X, y).transform(X)

    def inverse_transform(self, y):
        return self.classes_[np.asarray(y).reshape(-1)]


class SparseFeat:
    """inputs.py:148-210.  feat_size = cardinality + 1 (slot 0 = null/unknown, :166)."""

    def __init__(self, name, feat_size, weights=None, dtype=None, encoder=None, description=None):
        self.name = name
        self.dtype = dtype
        self.description = description
        self.encoder = encoder if encoder else ResilientLabelEncoder()
        self.feat_size = feat_size + 1
        self._weights = weights
        self._weights_cache = None

    @property
    def weights(self):
        """Manual per-class boosts added to the linear weights at predict time
        (inputs.py:170-182, layers.py:338-345)."""
        if self._weights:
            if self._weights_cache is None:
                ids = self.encoder.transform(list(self._weights.keys())) if self.encoder else np.array(
                    list(self._weights.keys()))
                w = np.zeros((self.feat_size,))
                for i, val in zip(np.asarray(ids).flatten(), self._weights.values()):
                    w[i] = val
                self._weights_cache = w
            return self._weights_cache
        return np.zeros((self.feat_size,))

    def set_weights(self, val):
        self._weights = val
        self._weights_cache = None

    def get_shape(self, for_tf=True):
        return None if for_tf else -1, 1

    def initialize(self, X):
        if self.encoder:
            self.encoder.fit(X)

    def __call__(self, x):
        if self.encoder:
            x = self.encoder.transform(x)
        return np.asarray(x).astype(np.int64).reshape(-1, 1)

    def decode(self, x):
        return self.encoder.inverse_transform(x) if self.encoder else x

    def __repr__(self):
        return f"SparseFeat({self.name}, {self.feat_size})"


class SparseValueFeat:
    """inputs.py:213-278: one (id, value) pair per example - the column holds pairs (a Series
    of 2-sequences or an [N,2] array).  Looked up as value * embedding[id] (layers.py:129-142;
    the bias lookup is NOT scaled, :136-140) and value * linear_w[id] (utils.py:70-71).

    Reference quirks (documented in DESIGN.md, not reproduced): the reference casts the whole
    [N,2] array to the feature dtype, int64 by default (:264), truncating the values, and then
    multiplies a float32 [B,1,D] tensor by the int64 [B] column (layers.py:142) - a dtype
    error in TF, and a mis-broadcast over D even with matching dtypes.  Here the value stays
    float32 and scales the example's row, which is what the docstring (:214-216) describes."""

    def __init__(self, name, feat_size, weights=None, dtype=None, encoder=None, description=None):
        self.name = name
        self.dtype = dtype
        self.description = description
        self.encoder = encoder if encoder else ResilientLabelEncoder()
        self.feat_size = feat_size + 1
        self._weights = weights
        self._weights_cache = None

    weights = SparseFeat.weights
    set_weights = SparseFeat.set_weights

    @staticmethod
    def _pairs(X):
        X = np.array(X.tolist(), dtype=object) if isinstance(X, pd.Series) else np.asarray(X, dtype=object)
        assert X.ndim == 2 and X.shape[1] == 2, "SparseValueFeat takes (id, value) pairs"  # :251,260
        return X

    def get_shape(self, for_tf=True):
        return None if for_tf else -1, 2

    def initialize(self, X):
        if self.encoder:
            self.encoder.fit(self._ids(self._pairs(X)))

    @staticmethod
    def _ids(X):
        ids = X[:, 0]
        try:  # homogeneous numeric ids keep their numeric dtype (the encoder branches on it)
            return ids.astype(np.int64) if all(isinstance(v, (int, np.integer)) for v in ids) else ids.astype(str)
        except (TypeError, ValueError):
            return ids.astype(str)

    def encode(self, x):
        """-> CSR with one id per example and vals = the float32 values."""
        X = self._pairs(x)
        ids = self._ids(X)
        if self.encoder:
            ids = self.encoder.transform(ids).reshape(-1)
        n = len(X)
        return CSR(np.arange(n + 1), np.asarray(ids, dtype=np.int64), X[:, 1].astype(np.float32))

    def __call__(self, x):
        c = self.encode(x)
        return np.stack([c.ids.astype(np.float64), c.vals.astype(np.float64)], axis=1)

    def decode(self, x):
        return self.encoder.inverse_transform(x) if self.encoder else x

    def __repr__(self):
        return f"SparseValueFeat({self.name}, {self.feat_size})"


class DenseFeat:
    """inputs.py:281-322: float32 cast -> sklearn scaler -> float32 [B,1].  (The reference's
    default argument `scaler=StandardScaler()` is one shared instance, :287; here every
    feature gets its own.)"""

    def __init__(self, name, weights=None, dtype=None, scaler=None, description=None):
        self.name = name
        self.dtype = dtype
        self.description = description
        self.scaler = scaler if scaler is not None else StandardScaler()
        self.feat_size = 1
        self._weights = weights

    @property
    def weights(self):
        return [self._weights if self._weights is not None else 0]

    def get_shape(self, for_tf=True):
        return None if for_tf else -1, 1

    def initialize(self, X):
        if self.scaler:
            self.scaler.fit(np.asarray(X).reshape(-1, 1))

    def __call__(self, x):
        x = np.array(x, dtype=np.float32)
        if self.scaler:
            x = self.scaler.transform(x.reshape(-1, 1))
        return np.asarray(x).astype(np.float32).reshape(-1, 1)

    def __repr__(self):
        return f"DenseFeat({self.name}, {self.feat_size})"


class CSR:
    """Ragged tag-id lists of a multi-valued feature: example b owns ids[offsets[b]:offsets[b+1]]
    (and, for a value feature, the per-id float32 weights vals)."""

    def __init__(self, offsets, ids, vals=None):
        self.offsets = np.asarray(offsets, dtype=np.int64)
        self.ids = np.asarray(ids, dtype=np.int64)
        self.vals = None if vals is None else np.asarray(vals, dtype=np.float32)

    @classmethod
    def from_lists(cls, lists):
        n = np.fromiter((len(x) for x in lists), dtype=np.int64, count=len(lists))
        offsets = np.concatenate(([0], np.cumsum(n)))
        ids = np.concatenate(lists) if len(lists) and offsets[-1] else np.zeros(0, np.int64)
        return cls(offsets, ids)

    def __len__(self):
        return len(self.offsets) - 1

    def slice(self, s, t):
        o = self.offsets[s: t + 1]
        return CSR(o - o[0], self.ids[o[0]: o[-1]], None if self.vals is None else self.vals[o[0]: o[-1]])

    def take(self, perm):
        n = self.offsets[1:] - self.offsets[:-1]
        starts = self.offsets[:-1][perm]
        lens = n[perm]
        offsets = np.concatenate(([0], np.cumsum(lens)))
        pos = np.repeat(starts - offsets[:-1], lens) + np.arange(offsets[-1])
        return CSR(offsets, self.ids[pos], None if self.vals is None else self.vals[pos])


class MultiValCsvFeat:
    """inputs.py:380-425: a '|'-separated tag list per example ("a|b|d"); tag -> 1-based id
    (tag_hash_table, :389), unknown tags -> 0; feat_size = len(tags) + 1.  Looked up with the
    sqrtn combiner (layers.py:144-169); the linear term sees its multi-hot counts with slot 0
    zeroed (utils.py:86-108)."""

    def __init__(self, name, tags=(), weights=None, dtype=None, description=None):
        self.name = name
        self.dtype = dtype
        self.description = description
        self.tags = tags
        self.tag_hash_table = dict((tag, i + 1) for i, tag in enumerate(self.tags))
        self.feat_size = len(self.tags) + 1
        self._weights = weights
        self._weights_cache = None

    def get_shape(self, for_tf=True):
        return None if for_tf else -1, 1

    def initialize(self, X):
        pass

    def __call__(self, x, training=True):
        return np.array(x).reshape(-1, 1)  # the reference hands the raw strings to the graph

    def encode(self, x):
        """-> CSR of tag ids; python str.split semantics (as tf.strings.split with a separator):
        the empty string is ONE empty token, which is unknown -> id 0."""
        table = self.tag_hash_table
        return CSR.from_lists([np.fromiter((table.get(t, 0) for t in str(v).split("|")), dtype=np.int64)
                               for v in np.asarray(x).reshape(-1)])

    def set_weights(self, val):
        self._weights = val
        self._weights_cache = None

    @property
    def weights(self):
        if self._weights:
            if self._weights_cache is None:
                w = np.zeros((self.feat_size,))
                for tag, weight in self._weights.items():
                    if tag in self.tag_hash_table:
                        w[self.tag_hash_table[tag]] = weight
                self._weights_cache = w
            return self._weights_cache
        return np.zeros((self.feat_size,))

    def __repr__(self):
        return f"MultiValCsvFeat({self.name}, {len(self.tags)})"


class FeatureDictionary(OrderedDict):
    """inputs.py:8-43.  Insertion order defines the field axis of E (and therefore the
    row order of the CIN filters): embedding_feats are the non-dense features in order."""

    @property
    def embedding_feats(self):
        return [f for f in self.values() if not isinstance(f, DenseFeat)]

    @property
    def sparse_feats(self):
        return [f for f in self.values() if isinstance(f, SparseFeat)]

    @property
    def dense_feats(self):
        return [f for f in self.values() if isinstance(f, DenseFeat)]

    @property
    def sparse_val_feats(self):
        return [f for f in self.values() if isinstance(f, SparseValueFeat)]

    @property
    def multi_val_csv_feats(self):
        return [f for f in self.values() if isinstance(f, MultiValCsvFeat)]

    @property
    def linear_feats(self):
        """get_linear_features (utils.py:27-36): sparse, value, multi-valued csv, then dense."""
        return self.sparse_feats + self.sparse_val_feats + self.multi_val_csv_feats + self.dense_feats

    def initialize(self, X):
        for feat in self.values():
            feat.initialize(X[feat.name])

    def check_supported(self):
        ok = (SparseFeat, SparseValueFeat, DenseFeat, MultiValCsvFeat)
        bad = [f for f in self.values() if not isinstance(f, ok)]
        if bad:
            raise NotImplementedError(
                f"features {[f.name for f in bad]}: SparseFeat, SparseValueFeat, DenseFeat and "
                "MultiValCsvFeat are on the HIP path (the reference itself raises NotImplementedError "
                "for MultiValSparseFeat lookups, utils.py:111-115, and for SequenceFeat, inputs.py:443)")


class DataInputs(dict):
    """inputs.py:46-90: name -> encoded array (+ 'y'), plus the packed arrays the kernels
    take: idx int64 [B,F] (embedding_feats order) and dense float32 [B,Dn]."""

    def load(self, feat_dict, X, y=None):
        for feat in feat_dict.values():
            self[feat.name] = feat(X[feat.name])
        if y is not None:
            self["y"] = np.asarray(y)
        sparse = feat_dict.embedding_feats
        dense = feat_dict.dense_feats
        n = len(X)
        # multi-valued features: tag ids as CSR; their idx column is a placeholder
        self.mv = {f.name: f.encode(X[f.name]) for f in sparse
                   if isinstance(f, (MultiValCsvFeat, SparseValueFeat))}
        cols = [np.zeros((n, 1), np.int64) if f.name in self.mv else self[f.name] for f in sparse]
        self.idx = np.concatenate(cols, axis=1) if sparse else np.zeros((n, 0), np.int64)
        self.dense = (np.concatenate([self[f.name] for f in dense], axis=1).astype(np.float32)
                      if dense else np.zeros((n, 0), np.float32))
        self.check_ranges(feat_dict)
        return self

    def check_ranges(self, feat_dict):
        """Every id must address a row of ITS feature: 0 <= id < feat_size.  The kernels compute
        field_off[f] + id without a bound (an undersized SparseFeat(name, feat_size) would read -
        and, in the optimizer, write - the next feature's rows or past the table); the reference
        fails loudly at the same spot (tf.nn.embedding_lookup raises InvalidArgument,
        layers.py:117-128).  One pass over arrays that are encoded once per fit()."""
        for j, f in enumerate(feat_dict.embedding_feats):
            ids = self.mv[f.name].ids if f.name in self.mv else self.idx[:, j]
            if ids.size == 0:
                continue
            lo, hi = int(ids.min()), int(ids.max())
            if lo < 0 or hi >= f.feat_size:
                raise ValueError(
                    f"feature {f.name!r}: encoded id {lo if lo < 0 else hi} outside [0, {f.feat_size}) - "
                    f"feat_size={f.feat_size - 1} is smaller than the fitted vocabulary")

    @property
    def y(self):
        return self["y"]

    def dense_inputs(self, feat_dict):
        return [self[f.name] for f in feat_dict.dense_feats]

    def sparse_inputs(self, feat_dict):
        return [self[f.name] for f in feat_dict.sparse_feats]

    def embedding_inputs(self, feat_dict):
        return [self[f.name] for f in feat_dict.embedding_feats]
