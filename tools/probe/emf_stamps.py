"""Phase times of embed_mlp_fwd_kernel (lane 0 of every wave) from a -DRM_EMF_STAMP build
(RECMAN_HIP_LIB=build/librecman_est.so): DeepFM configs[1], 60 eager steps, stamps of the last launch."""
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
assert lib.rm_debug_emf_stamps(buf, 8 * n) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.float64) / 100.0
t0 = s[:, 4].min()
end = s[:, 4] - t0 + s[:, 0] + s[:, 1] + s[:, 2]
print(f"per wave, us: prologue {s[:, 0].mean():.2f}  chunk loop {s[:, 1].mean():.2f} (of it waiting for row loads "
      f"{s[:, 3].mean():.2f})  FM finalize + MLP epilogue + head {s[:, 2].mean():.2f}; wave end mean {end.mean():.2f} "
      f"max {end.max():.2f}; start spread {s[:, 4].max() - t0:.2f}")
