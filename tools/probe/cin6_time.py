"""rm_cin_layer_fwd6 (bf16 pipe, split operands) against rm_cin_layer_fwd (f32 MFMA): time, agreement, error vs float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from recman_amd import ops


def timed(fn, it=10):
    for _ in range(2):
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
for (B, m, H, N, D, pf) in [(300, 26, 64, 128, 16, 64), (65536, 26, 64, 128, 16, 64), (8192, 26, 64, 128, 64, 64),
                            (1000, 7, 32, 48, 32, 0)]:
    X0 = torch.randn(B, m, D, generator=g, device="cuda")
    Xk = torch.randn(B, H + 3, D, generator=g, device="cuda")
    W = torch.randn(m * H, N, generator=g, device="cuda") / (m * H) ** 0.5
    bias = torch.randn(N, generator=g, device="cuda")
    o6, o32 = torch.empty(B, N, D, device="cuda"), torch.empty(B, N, D, device="cuda")
    p6, p32 = torch.zeros(B, N - pf + 5, device="cuda"), torch.zeros(B, N - pf + 5, device="cuda")
    ws = torch.empty(ops.cin_filter_workspace(m, H, N), device="cuda")
    ws6 = torch.empty(ops.cin_filter_workspace6(m, H, N, D), device="cuda")
    f6 = lambda: ops.cin_layer_fwd(X0, Xk, H, W, bias, "leaky_relu", o6, ws, pooled=p6, pool_col0=2, pool_from=pf, ws6=ws6)
    f32 = lambda: ops.cin_layer_fwd(X0, Xk, H, W, bias, "leaky_relu", o32, ws, pooled=p32, pool_col0=2, pool_from=pf)
    assert f6() is True
    f32()
    torch.cuda.synchronize()
    n = min(B, 64)
    Z = (X0[:n, :, None, :] * Xk[:n, None, :H, :]).reshape(n, m * H, D)           # fp32 products, as the reference
    ref = torch.einsum("bkd,kn->bnd", Z.double(), W.double()) + bias.double()[None, :, None]
    ref = torch.nn.functional.leaky_relu(ref, 0.2)
    e6 = float((o6[:n].double() - ref).abs().max())
    e32 = float((o32[:n].double() - ref).abs().max())
    ep = float((p6 - p32).abs().max())
    print(f"B={B} m={m} H={H} N={N} D={D}: bf16x6 {timed(f6):8.1f} us (err {e6:.2e}), f32 {timed(f32):8.1f} us (err {e32:.2e}); "
          f"|out6 - out32| {float((o6 - o32).abs().max()):.2e}, pooled diff {ep:.2e}", flush=True)
