"""The bf16x6 TN GEMM (rm_dense_wgrad6) against the f32-MFMA one (rm_dense_wgrad): time and error vs float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from recman_amd import ops


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
SH = [(4096, 416, 13, 400), (131072, 416, 13, 400), (131072, 400, 0, 400)]
if os.environ.get("TN6_ONE"):
    SH = SH[1:2]
for (M, K1, K2, N) in SH:
    a1 = torch.randn(M, K1, generator=g, device="cuda")
    a2 = torch.randn(M, K2, generator=g, device="cuda") if K2 else None
    G = torch.randn(M, N, generator=g, device="cuda") * 0.1
    K = K1 + K2
    d6, d32 = torch.empty(K, N, device="cuda"), torch.empty(K, N, device="cuda")
    b6, b32 = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    ws6 = torch.zeros(ops.dense_wgrad6_workspace(K, N, M), device="cuda")
    ws32 = torch.zeros(ops.dense_wgrad_workspace(K, N, M), device="cuda")
    f6 = lambda: ops.dense_wgrad(a1, a2, G, d6, ws32, db=b6, ws6=ws6)
    f32 = lambda: ops.dense_wgrad(a1, a2, G, d32, ws32, db=b32)
    f6(); f32()
    torch.cuda.synchronize()
    x = torch.cat([a1] + ([a2] if K2 else []), 1)
    ref = (x.double().t() @ G.double())
    e6 = float((d6.double() - ref).abs().max())
    e32 = float((d32.double() - ref).abs().max())
    eb = float((b6.double() - G.double().sum(0)).abs().max())
    print(f"M={M} K={K1}+{K2} N={N}: bf16x6 {timed(f6):7.1f} us (max err {e6:.2e}, db err {eb:.2e}), f32 MFMA "
          f"{timed(f32):7.1f} us (max err {e32:.2e}); |ref| max {float(ref.abs().max()):.1f}", flush=True)
