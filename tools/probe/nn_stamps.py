"""Per-block phase times of dense_nn_kernel from a -DRM_NN_STAMP build (RECMAN_HIP_LIB=build/librecman_stamp.so):
s_memrealtime stamps (100 MHz) at block start / after the prologue / after the chunk loop / after the epilogue."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recman_amd import _lib, ops  # noqa: E402

B, H = int(os.environ.get("DENSE_B", 131072)), 400
dev = "cuda"
h1, h2 = torch.randn(B, H, device=dev), torch.empty(B, H, device=dev)
W2, b1 = torch.randn(H, H, device=dev) * 0.05, torch.zeros(H, device=dev)
fws = torch.empty(ops.dense_filter_workspace(H, 448), device=dev)
for _ in range(int(os.environ.get("DENSE_WARM", 15))):
    ops.dense_fwd(h1, None, W2, h2, fws, bias=b1, act="relu")
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
n = min(8192, (B + 127) // 128 * 2)
buf = (ctypes.c_ulonglong * (8 * n))()
assert lib.rm_debug_nn_stamps(buf, 8 * n) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
t0 = s[:, 0].min()
st, pro, loop, epi = (s[:, 0] - t0) / 100.0, (s[:, 1] - s[:, 0]) / 100.0, (s[:, 2] - s[:, 1]) / 100.0, (s[:, 3] - s[:, 2]) / 100.0
end = (s[:, 3] - t0) / 100.0
print(f"kernel span {end.max():.1f} us over {n} blocks")
for g in (0, 1):
    m = s[:, 5] == g
    print(f"group {g}: prologue {pro[m].mean():.2f} us  loop {loop[m].mean():.2f} (min {loop[m].min():.2f} max {loop[m].max():.2f})  "
          f"epilogue {epi[m].mean():.2f} (max {epi[m].max():.2f})  block {(end - st)[m].mean():.2f}")
ghz = s[:, 7] / np.maximum(s[:, 2] - s[:, 1], 1) * 0.1
print(f"shader clock over the chunk loops: median {np.median(ghz):.3f} GHz (min {ghz.min():.3f} max {ghz.max():.3f})")
print("clock by block start time: " + "  ".join(
    f"[{lo:.0f}-{hi:.0f} us] {np.median(ghz[(st >= lo) & (st < hi)]):.3f}" for lo, hi in
    [(q * end.max() / 8, (q + 1) * end.max() / 8) for q in range(8)] if ((st >= lo) & (st < hi)).any()))
hw, xcc = s[:, 4], s[:, 6] & 15
cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
order = np.argsort(st)
print("first 12 blocks by start: (block, start us, end us, cu, group)")
for b in order[:12]:
    print(f"  {b:5d} {st[b]:8.2f} {end[b]:8.2f}  cu {cu[b]:5d}  g {s[b, 5]}")
# blocks per CU and their overlap: for one CU list (start, end, group)
ucu, cnt = np.unique(cu, return_counts=True)
print(f"{len(ucu)} distinct CU ids; blocks per CU min {cnt.min()} max {cnt.max()}")
c0 = ucu[0]
for b in np.where(cu == c0)[0][np.argsort(st[cu == c0])]:
    print(f"  cu {c0}: block {b:5d} g {s[b, 5]} simd {(hw[b] >> 4) & 3}  {st[b]:8.2f} -> {end[b]:8.2f}  (pro {pro[b]:.2f} loop {loop[b]:.2f} epi {epi[b]:.2f})")
