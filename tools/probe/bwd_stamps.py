"""Phase times of mlp_bwd_kernel (wave 0 of every block) from a -DRM_BWD_STAMP build
(RECMAN_HIP_LIB=build/librecman_bst.so): DeepFM configs[1], 60 eager steps, stamps of the last launch."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from recman_amd import _lib  # noqa: E402

a = bench.parse(["--workload", "deepfm", "--only", "--no-pmc", "--no-optimizer", "--no-cpu-baseline"])
dev = torch.device("cuda", 0)
w = bench.WORKLOADS["deepfm"]
engine, idx, dense, y, hp = bench.make_engine(a, w, w["B"], w["V"], dev, 0, 1, False)
for _ in range(60):
    engine.fwd_bwd(idx, dense, y)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
n = 256 * 8
buf = (ctypes.c_ulonglong * (8 * n))()
assert lib.rm_debug_bwd_stamps(buf, 8 * n) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 8).astype(np.float64) / 100.0
t0 = s[:, :, 5].min()
for wv in range(8):
    q = s[:, wv, :]
    print(f"wave {wv}: prologue {q[:, 0].mean():5.2f}  tile loop {q[:, 1].mean():6.2f} (staging {q[:, 3].mean():5.2f} + k-tiles "
          f"{q[:, 4].mean():6.2f})  final store {q[:, 2].mean():5.2f}  end at {(q[:, 5] - t0 + q[:, 0] + q[:, 1] + q[:, 2]).mean():6.2f} us "
          f"(max over blocks {(q[:, 5] - t0 + q[:, 0] + q[:, 1] + q[:, 2]).max():6.2f})")
