"""CPU: the product's vectorised encoders against the oracle's restatement of the
reference encoders (bit-exact indices) and against the committed ml-100k golden slice."""
import os

import numpy as np
import pandas as pd
import pytest
from sklearn.preprocessing import MinMaxScaler, StandardScaler

from oracle import inputs_ref as R
from recman_amd.th.inputs import (DataInputs, DenseFeat, FeatureDictionary, ResilientLabelEncoder,
                                  SparseFeat)

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "ml100k_slice.npz"))


def test_survey_encoder_vectors():
    # strings fit on the first four -> [2,1,3,2,0]; ints -> zeros (strict)
    s = pd.Series(["b", "a", "c", "b", "zz"])
    ref = R.RefLabelEncoder().fit(s[:4])
    assert ref.transform(s).reshape(-1).tolist() == [2, 1, 3, 2, 0]
    enc = ResilientLabelEncoder().fit(s[:4])
    assert enc.transform(s).reshape(-1).tolist() == [2, 1, 3, 2, 0]
    ints = pd.Series([5, 3, 9, 5, 77])
    with pytest.warns(FutureWarning):
        assert R.RefLabelEncoder().fit(ints[:4]).transform(ints).reshape(-1).tolist() == [0] * 5
    assert ResilientLabelEncoder(strict_reference=True).fit(ints[:4]).transform(ints).reshape(-1).tolist() == [0] * 5
    assert ResilientLabelEncoder().fit(ints[:4]).transform(ints).reshape(-1).tolist() == [2, 1, 3, 2, 0]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_string_columns_bit_exact_vs_reference_restatement(seed):
    rng = np.random.default_rng(seed)
    vocab = np.array(["u%d" % i for i in range(300)] + ["-----", "Z", "a b", "0", "~x"], dtype=object)
    train = pd.Series(rng.choice(vocab, 2000))
    test = pd.Series(rng.choice(np.concatenate([vocab, np.array(["new1", "new2"], dtype=object)]), 500))
    ref = R.RefLabelEncoder().fit(train)
    enc = ResilientLabelEncoder().fit(train)
    assert np.array_equal(ref.transform(test), enc.transform(test))
    assert np.array_equal(ref.transform(train), enc.transform(train))
    assert enc.transform(test).dtype == np.int64


def test_golden_ml100k_slice_indices_and_dense():
    for c in ["gender", "occupation", "zip"]:  # string columns: reference behaviour reproduced
        raw = pd.Series(GOLD["raw_" + c].astype(object))
        f = SparseFeat(c, feat_size=len(np.unique(raw)))
        f.initialize(raw)
        assert np.array_equal(f(raw).reshape(-1), GOLD["ref_idx_" + c]), c
        assert f(raw).dtype == np.int64 and f(raw).shape == (1024, 1)
    for c in ["user_id", "item_id"]:  # integer ids: all-zero under the reference (quirk a2)
        raw = pd.Series(GOLD["raw_" + c])
        assert not GOLD["ref_idx_" + c].any()
        strict = SparseFeat(c, 10, encoder=ResilientLabelEncoder(strict_reference=True))
        strict.initialize(raw)
        assert not strict(raw).any()
        useful = SparseFeat(c, 10)
        useful.initialize(raw)
        got = useful(raw).reshape(-1)
        # by-value encoding == rank of the id + 1; cast-to-str orders lexicographically instead
        assert np.array_equal(got, np.searchsorted(np.unique(raw.values), raw.values) + 1)
        as_str = SparseFeat(c, 10)
        as_str.initialize(raw.astype(str))
        assert np.array_equal(as_str(raw.astype(str)).reshape(-1), GOLD["str_idx_" + c])
    for c in ["timestamp", "age"]:
        raw = pd.Series(GOLD["raw_" + c])
        f = DenseFeat(c, scaler=MinMaxScaler())
        f.initialize(raw)
        got = f(raw)
        assert got.dtype == np.float32 and got.shape == (1024, 1)
        assert np.array_equal(got.reshape(-1), GOLD["ref_dense_" + c]), c


def test_dense_feat_standard_scaler_matches_reference_restatement():
    rng = np.random.default_rng(3)
    x = pd.Series(rng.normal(5, 3, 777))
    f = DenseFeat("x")
    f.initialize(x)
    sc = R.dense_feat_fit(x, StandardScaler())
    assert np.array_equal(f(x), R.dense_feat_encode(sc, x))
    # every feature owns its scaler (the reference shares one default instance, inputs.py:287)
    assert DenseFeat("a").scaler is not DenseFeat("b").scaler


def test_feature_dictionary_order_and_packing():
    fd = FeatureDictionary()
    fd["d0"] = DenseFeat("d0")
    fd["s1"] = SparseFeat("s1", 3)
    fd["s0"] = SparseFeat("s0", 2)
    fd["d1"] = DenseFeat("d1")
    df = pd.DataFrame({"s0": ["a", "b", "a"], "s1": ["x", "y", "z"], "d0": [1., 2., 3.], "d1": [0., 0., 1.]})
    fd.initialize(df)
    assert [f.name for f in fd.embedding_feats] == ["s1", "s0"]  # insertion order (inputs.py:13-15)
    inp = DataInputs().load(fd, df, np.array([1, 0, 1]))
    assert inp.idx.shape == (3, 2) and inp.idx.dtype == np.int64
    assert inp.idx[:, 0].tolist() == [1, 2, 3] and inp.idx[:, 1].tolist() == [1, 2, 1]
    assert inp.dense.shape == (3, 2) and inp.dense.dtype == np.float32
    assert [f.feat_size for f in fd.sparse_feats] == [4, 3]  # + null slot (inputs.py:166)
    assert inp.y.tolist() == [1, 0, 1]


def test_manual_weights_vector():
    f = SparseFeat("c", 3, weights={"Outdoor": -5})
    f.initialize(pd.Series(["Outdoor", "Rest", "Treadmill"]))
    assert f.weights.tolist() == [0, -5, 0, 0]
    f.set_weights({"Rest": 2.0, "unseen": 9.0})
    assert f.weights.tolist() == [9.0, 0, 2.0, 0]  # unseen keys land on the null slot, as in the reference


def test_multi_val_csv_feat_encoding_and_weights():
    from recman_amd.th.inputs import CSR, MultiValCsvFeat

    f = MultiValCsvFeat("h", tags=("a", "b", "c", "d"))
    assert f.feat_size == 5 and f.tag_hash_table == {"a": 1, "b": 2, "c": 3, "d": 4}  # inputs.py:389-391
    c = f.encode(np.array(["a|b|d", "b|c", "", "zz|a"]))
    assert c.offsets.tolist() == [0, 3, 5, 6, 8]
    assert c.ids.tolist() == [1, 2, 4, 2, 3, 0, 0, 1]  # unknown tag and the empty string -> 0
    assert f(np.array(["a|b", "c"])).shape == (2, 1)   # the reference passes the raw strings on
    f.set_weights({"b": -5, "nope": 3})
    assert f.weights.tolist() == [0, 0, -5, 0, 0]
    t = c.take(np.array([3, 0, 2, 1]))
    assert t.offsets.tolist() == [0, 2, 5, 6, 8] and t.ids.tolist() == [0, 1, 1, 2, 4, 0, 2, 3]
    s = c.slice(1, 3)
    assert s.offsets.tolist() == [0, 2, 3] and s.ids.tolist() == [2, 3, 0]


def test_linear_feature_order_with_multi_val():
    from recman_amd.th.inputs import MultiValCsvFeat

    fd = FeatureDictionary()
    fd["s0"] = SparseFeat("s0", 2)
    fd["mv"] = MultiValCsvFeat("mv", tags=("x", "y"))
    fd["d0"] = DenseFeat("d0")
    fd["s1"] = SparseFeat("s1", 3)
    assert [f.name for f in fd.embedding_feats] == ["s0", "mv", "s1"]      # field axis of E
    assert [f.name for f in fd.linear_feats] == ["s0", "s1", "mv", "d0"]    # utils.py:31-36


def test_sparse_value_feat_encodes_pairs():
    """SparseValueFeat (inputs.py:213-278): (id, value) pairs -> encoded id + float32 value; the
    one-id CSR the scratch-row kernel takes; linear-feature order sparse, value, multi, dense."""
    import pandas as pd

    from recman_amd.th import (DataInputs, DenseFeat, FeatureDictionary, MultiValCsvFeat, SparseFeat,
                               SparseValueFeat)

    df = pd.DataFrame({
        "tagw": [("b", 0.5), ("a", 2.0), ("zz", -1.0), ("b", 0.0)],
        "city": ["x", "y", "x", "q"],
        "g": ["p|q", "", "q", "p"],
        "age": [1.0, 2.0, 3.0, 4.0],
    })
    fd = FeatureDictionary()
    fd["tagw"] = SparseValueFeat("tagw", feat_size=3)
    fd["g"] = MultiValCsvFeat("g", tags=("p", "q"))
    fd["age"] = DenseFeat("age")
    fd["city"] = SparseFeat("city", feat_size=3)
    fd.initialize(df)
    assert [f.name for f in fd.linear_feats] == ["city", "tagw", "g", "age"]
    assert [f.name for f in fd.embedding_feats] == ["tagw", "g", "city"]
    inp = DataInputs().load(fd, df)
    c = inp.mv["tagw"]
    assert c.ids.tolist() == [2, 1, 3, 2] and c.offsets.tolist() == [0, 1, 2, 3, 4]
    assert c.vals.dtype == np.float32 and c.vals.tolist() == [0.5, 2.0, -1.0, 0.0]
    assert inp["tagw"].shape == (4, 2) and inp["tagw"][:, 0].tolist() == [2, 1, 3, 2]
    assert inp.idx.shape == (4, 3) and inp.idx[:, 0].tolist() == [0, 0, 0, 0]  # placeholder column
    s = c.take(np.array([3, 0])).slice(0, 2)
    assert s.ids.tolist() == [2, 2] and s.vals.tolist() == [0.0, 0.5]
    unseen = fd["tagw"].encode(pd.Series([("nope", 1.5)]))
    assert unseen.ids.tolist() == [0] and unseen.vals.tolist() == [1.5]
    fd["tagw"].set_weights({"a": 3.0})
    assert fd["tagw"].weights.tolist() == [0.0, 3.0, 0.0, 0.0]


def test_ids_outside_feat_size_raise():
    """An undersized feat_size must fail on the host (the reference's tf.nn.embedding_lookup raises
    InvalidArgument): the kernels address field_off[f] + id without a bound."""
    from recman_amd.th.inputs import MultiValCsvFeat

    df = pd.DataFrame({"s": ["a", "b", "c", "d"], "t": ["x", "x", "y", "y"]})
    fd = FeatureDictionary()
    fd["s"] = SparseFeat("s", 2)   # 4 fitted classes need feat_size >= 4
    fd["t"] = SparseFeat("t", 2)
    fd.initialize(df)
    with pytest.raises(ValueError, match="feature 's'"):
        DataInputs().load(fd, df)
    fd["s"] = SparseFeat("s", 4)
    fd.initialize(df)
    assert DataInputs().load(fd, df).idx[:, 0].tolist() == [1, 2, 3, 4]
    # an encoder-less feature handing in raw ids (negative / too large) and CSR tag ids
    raw = SparseFeat("r", 3, encoder=None)
    raw.encoder = None
    fd2 = FeatureDictionary()
    fd2["r"] = raw
    with pytest.raises(ValueError, match="feature 'r'"):
        DataInputs().load(fd2, pd.DataFrame({"r": [0, 1, -1]}))
    with pytest.raises(ValueError, match="feature 'r'"):
        DataInputs().load(fd2, pd.DataFrame({"r": [0, 4, 1]}))
    assert DataInputs().load(fd2, pd.DataFrame({"r": [0, 3, 1]})).idx.reshape(-1).tolist() == [0, 3, 1]
    mv = MultiValCsvFeat("g", tags=("p", "q"))
    mv.tag_hash_table["bad"] = 7   # a corrupted tag table: id beyond feat_size
    fd3 = FeatureDictionary()
    fd3["g"] = mv
    with pytest.raises(ValueError, match="feature 'g'"):
        DataInputs().load(fd3, pd.DataFrame({"g": ["p|bad"]}))
