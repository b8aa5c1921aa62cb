"""Per-wave phase times of deepfm_step_kernel from a -DRM_STEP_STAMP build
(python -m recman_amd.build --out build/librecman_sst.so --only step -DRM_STEP_STAMP; RECMAN_HIP_LIB=build/librecman_sst.so):
DeepFM configs[1], eager steps, stamps of the last launch (shader-clock ticks -> us at the measured clock)."""
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
w = dict(w, hp=dict(w["hp"], step_fusion=True))
engine, idx, dense, y, hp = bench.make_engine(a, w, w["B"], w["V"], dev, 0, 1, False)
for _ in range(200):
    engine.fwd_bwd(idx, dense, y)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(50):
    engine.fwd_bwd(idx, dense, y)
ev[1].record()
torch.cuda.synchronize()
print(f"eager step {ev[0].elapsed_time(ev[1]) / 50 * 1e3:.1f} us")
lib = ctypes.CDLL(_lib.LIB_PATH)
n = 256 * 8
buf = (ctypes.c_ulonglong * (8 * n))()
assert lib.rm_debug_step_stamps(buf, 8 * n) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 8).astype(np.float64)
names_w = ["pro+epilogue", "bwd reads", "dma issue", "row wait", "forward", "barrier", "bwd mfma+store", "total"]
names_h = ["pro+epilogue", "partial+FM", "L1+loss", "chain+publish", "small grads", "barrier", "prefetch", "total"]
print("ticks are shader-clock cycles (s_memtime); mean over the 256 blocks")
for wv in range(8):
    q = s[:, wv, :].mean(0)
    nm = names_h if wv == 7 else names_w
    print(f"wave {wv}: " + "  ".join(f"{nm[i]} {q[i]:8.0f}" for i in range(8) if nm[i] != "-"))
