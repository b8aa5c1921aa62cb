"""Per-block durations of cin_dw_kernel by k'-tile group (RM_CIN_STAMP build): layer 1 of configs[2] (H = 64) and
the symmetric layer 0, 12 back-to-back backward calls each, stamps of the last dW launch."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recman_amd import _lib, ops  # noqa: E402

B, m, D = 65536, 26, 16
dev = "cuda"
X0 = torch.randn(B, m, D, device=dev) * 0.1
lib = ctypes.CDLL(_lib.LIB_PATH)
for (H, N, first) in ((64, 128, False), (26, 128, True)):
    Xk = X0 if first else torch.randn(B, 128, D, device=dev) * 0.1
    W = torch.randn(m * H, N, device=dev) * 0.05
    out = torch.randn(B, N, D, device=dev)
    g = torch.randn(B, device=dev)
    dX0 = torch.zeros(B, m, D, device=dev)
    dXk = None if first else torch.empty(B, H, D, device=dev)
    dW, db = torch.empty(m * H, N, device=dev), torch.empty(N, device=dev)
    ws = torch.empty(ops.cin_bwd_workspace(B, m, H, N, D), device=dev)
    dh, cw = torch.randn(B, 64, D, device=dev), torch.randn(64, device=dev)
    for _ in range(12):
        ops.cin_layer_bwd(X0, Xk, H, W, "leaky_relu", out, g, dX0, dW, db, ws, xk_is_x0=first, d_hidden=dh,
                          cin_w_direct=cw, pool_from=64, accumulate_dx0=True, dXk=dXk)
    torch.cuda.synchronize()
    n = 1024
    buf = (ctypes.c_ulonglong * (4 * n))()
    assert lib.rm_debug_cin_dw_stamps(buf, 4 * n) == 0
    s = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
    s = s[s[:, 1] > 0]
    t0 = s[:, 0].min()
    dur, end = (s[:, 1] - s[:, 0]) / 100.0, (s[:, 1] - t0) / 100.0
    print(f"H = {H}{' (symmetric first layer)' if first else ''}: {len(s)} blocks, kernel span {end.max():.1f} us")
    for grp in sorted(set(s[:, 2].tolist())):
        mk = s[:, 2] == grp
        print(f"  group {grp}: {mk.sum():3d} blocks x {s[mk, 3][0]} chunks, duration mean {dur[mk].mean():7.1f} max {dur[mk].max():7.1f} us, "
              f"last end {end[mk].max():7.1f}")
