#!/usr/bin/env python
"""Times the CIN kernels alone at the config-3 shape (hipEvents), per layer."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recman_amd import ops
B, m, D = 65536, 26, 16
dev = "cuda"
X0 = torch.randn(B, m, D, device=dev) * 0.1
def timeit(fn, n=5):
    for _ in range(2): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / n
for (H, N, first) in ((26, 128, True), (64, 128, False)):
    Xk = X0 if first else torch.randn(B, 128, D, device=dev) * 0.1
    W = torch.randn(m * H, N, device=dev) * 0.05; bias = torch.zeros(N, device=dev)
    out = torch.empty(B, N, D, device=dev); pooled = torch.empty(B, 192, device=dev)
    fws = torch.empty(ops.cin_filter_workspace(m, H, N), device=dev)
    flop = 2.0 * B * D * m * H * N
    t = timeit(lambda: ops.cin_layer_fwd(X0, Xk, H, W, bias, "leaky_relu", out, fws, pooled=pooled, pool_col0=0, pool_from=64))
    print(f"fwd  m={m} H={H} N={N}: {t:7.3f} ms  {flop/t/1e9:6.1f} TFLOP/s  {flop/t/1e9/157.3*100:5.1f}% of f32 MFMA peak")
    g = torch.randn(B, device=dev); dX0 = torch.zeros(B, m, D, device=dev)
    dXk = None if first else torch.empty(B, H, D, device=dev)
    dW = torch.empty(m * H, N, device=dev); db = torch.empty(N, device=dev)
    ws = torch.empty(ops.cin_bwd_workspace(B, m, H, N, D), device=dev)
    dh = torch.randn(B, 64, D, device=dev); cw = torch.randn(64, device=dev)
    t = timeit(lambda: ops.cin_layer_bwd(X0, Xk, H, W, "leaky_relu", out, g, dX0, dW, db, ws, xk_is_x0=first, d_hidden=dh, cin_w_direct=cw, pool_from=64, accumulate_dx0=True, dXk=dXk))
    print(f"bwd  m={m} H={H} N={N}: {t:7.3f} ms  {2*flop/t/1e9:6.1f} TFLOP/s  {2*flop/t/1e9/157.3*100:5.1f}% (dM + dX + dW + reduce)")
