"""In-kernel clock and phase times of cin_fwd_kernel from a -DRM_CIN_STAMP build (RECMAN_HIP_LIB=...): layer 1 of
configs[2] launched WARM times back to back (CIN_WARM, default 20), stamps of the last launch."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recman_amd import _lib, ops  # noqa: E402

B, m, D, H, N = 65536, 26, 16, 64, 128
dev = "cuda"
X0 = torch.randn(B, m, D, device=dev) * 0.1
Xk = torch.randn(B, 128, D, device=dev) * 0.1
W = torch.randn(m * H, N, device=dev) * 0.05
bias = torch.zeros(N, device=dev)
out, pooled = torch.empty(B, N, D, device=dev), torch.empty(B, 192, device=dev)
fws = torch.empty(ops.cin_filter_workspace(m, H, N), device=dev)
warm = int(os.environ.get("CIN_WARM", 20))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(warm):
    if i == warm - 1:
        e0.record()
    ops.cin_layer_fwd(X0, Xk, H, W, bias, "leaky_relu", out, fws, pooled=pooled, pool_col0=0, pool_from=64)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
flop = 2.0 * B * D * m * H * N
print(f"last launch {ms:.3f} ms  {flop / ms / 1e9:.1f} TFLOP/s = {flop / ms / 1e9 / 157.3:.3f} of peak")
lib = ctypes.CDLL(_lib.LIB_PATH)
n = B * D // 256
buf = (ctypes.c_ulonglong * (4 * n))()
assert lib.rm_debug_cin_stamps(buf, 4 * n) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.float64)
ghz = s[:, 0] / s[:, 1] * 0.1
print(f"clock in the chunk loop: median {np.median(ghz):.3f} GHz ({ghz.min():.3f}-{ghz.max():.3f})")
print(f"per block: prologue {np.mean(s[:, 3]) / 100:.2f} us, chunk loop {np.mean(s[:, 1]) / 100:.2f} us, "
      f"epilogue {np.mean(s[:, 2] - s[:, 1] - s[:, 3]) / 100:.2f} us, total {np.mean(s[:, 2]) / 100:.2f} us")
nch = (m * H + 31) // 32
cyc = nch * 16 * 4 * 64 * 2  # MFMA cycles per SIMD per block: chunks x k-steps x NT x 64 x two waves
print(f"MFMA-busy inside the chunk loop: {cyc / np.median(s[:, 0]):.3f} ({cyc} cycles of {np.median(s[:, 0]):.0f})")
