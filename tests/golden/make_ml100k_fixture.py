"""Generates tests/golden/ml100k_slice.npz (run in the build container only; the GPU box
never sees /root/reference).  Inputs: the reference's own loader
recman/examples/datasets/ml_100k.py:get_data on data/ml-100k (imports without TensorFlow).
Outputs: the first 1024 joined training rows (raw columns) + their encodings by the
oracle's restatement of the reference encoders (oracle/inputs_ref.py), fitted on the slice,
+ labels by the rule of recman/examples/utils.py:14-17 (rating >= 4)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from recman.examples.datasets.ml_100k import get_data  # noqa: E402  (reference, data loading only)
from sklearn.preprocessing import MinMaxScaler  # noqa: E402

from oracle import inputs_ref as R  # noqa: E402

df, _, _domains = get_data("/root/reference/data")
df = df.iloc[:1024].reset_index(drop=True)
out = {}
sparse = ["user_id", "item_id", "gender", "occupation", "zip"]
dense = ["timestamp", "age"]
for c in sparse:
    col = df[c]
    out["raw_" + c] = col.values.astype(str) if col.dtype == object else col.values
    enc = R.RefLabelEncoder().fit(col)
    out["ref_idx_" + c] = R.sparse_feat_encode(enc, col).reshape(-1)       # reference behaviour
    enc_s = R.RefLabelEncoder().fit(col.astype(str))
    out["str_idx_" + c] = R.sparse_feat_encode(enc_s, col.astype(str)).reshape(-1)  # ids as strings
for c in dense:
    out["raw_" + c] = df[c].values
    sc = R.dense_feat_fit(df[c], MinMaxScaler())  # recman/examples/utils.py:57-66
    out["ref_dense_" + c] = R.dense_feat_encode(sc, df[c]).reshape(-1)
out["raw_genres"] = df["genres"].values.astype(str)  # "Action|Comedy" strings from the reference loader
out["genre_tags"] = np.array(_domains["genres"]).astype(str)
out["label"] = (df["rating"].values >= 4).astype(np.int64)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ml100k_slice.npz"), **out)
print({k: (v.shape, v.dtype) for k, v in out.items()})
