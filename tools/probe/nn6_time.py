"""The bf16x6 NN GEMM (csrc/gemm6.hip, experimental entry rm_dense_fwd6) against the f32-MFMA kernel (rm_dense_fwd):
time and error against float64, at DCN's shapes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from recman_amd import ops, _lib

L = _lib.lib()
P, I64, CI = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
L.rm_dense_fwd6.argtypes = [P, I64, CI, P, I64, CI, P, I64, CI, CI, P, CI, CI, P, I64, I64, P, I64, P, P, P, P, P]
L.rm_dense_fwd6.restype = CI
L.rm_dense6_workspace.argtypes = [CI, CI, I64]
L.rm_dense6_workspace.restype = I64


def fwd6(a1, a2, W, out, ws, transposed=False, bias=None, epi=0, act=1, aux1=None):
    K1, K2 = a1.shape[1], (a2.shape[1] if a2 is not None else 0)
    N = W.shape[0] if transposed else W.shape[1]
    rc = L.rm_dense_fwd6(a1.data_ptr(), a1.stride(0), K1, a2.data_ptr() if a2 is not None else None,
                         a2.stride(0) if a2 is not None else 0, K2, W.data_ptr(), W.stride(0), int(transposed), N,
                         bias.data_ptr() if bias is not None else None, epi, act,
                         aux1.data_ptr() if aux1 is not None else None, aux1.stride(0) if aux1 is not None else 0,
                         a1.shape[0], out.data_ptr(), out.stride(0), None, None, None, ws.data_ptr(),
                         torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _lib.last_error() if hasattr(_lib, "last_error") else rc


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
SHAPES = [(4096, 416, 13, 400, False), (131072, 416, 13, 400, False), (131072, 400, 0, 400, False),
          (131072, 400, 0, 416, True)]
if os.environ.get("NN6_ONE"):
    SHAPES = SHAPES[1:2]
for (M, K1, K2, N, tr) in SHAPES:
    a1 = torch.randn(M, K1, generator=g, device="cuda")
    a2 = torch.randn(M, K2, generator=g, device="cuda") if K2 else None
    K = K1 + K2
    W = torch.randn((N, K) if tr else (K, N), generator=g, device="cuda") * 0.05
    bias = torch.randn(N, generator=g, device="cuda")
    out6, out32 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ws6 = torch.zeros(L.rm_dense6_workspace(K, N, M), device="cuda")
    ws32 = torch.zeros(ops.dense_filter_workspace(K, N), device="cuda")
    f6 = lambda: fwd6(a1, a2, W, out6, ws6, transposed=tr, bias=bias)
    f32 = lambda: ops.dense_fwd(a1, a2, W, out32, ws32, transposed=tr, bias=bias, act="relu")
    f6(); f32()
    torch.cuda.synchronize()
    n = min(M, 4096)
    x = torch.cat([a1[:n]] + ([a2[:n]] if K2 else []), 1).double()
    ref = torch.relu(x @ (W.double().t() if tr else W.double()) + bias.double())
    e6 = float((out6[:n].double() - ref).abs().max())
    e32 = float((out32[:n].double() - ref).abs().max())
    scale = float(ref.abs().max())
    print(f"M={M} K={K1}+{K2} N={N} tr={tr}: bf16x6 {timed(f6):7.1f} us (max err {e6:.2e}), f32 MFMA "
          f"{timed(f32):7.1f} us (max err {e32:.2e}); |ref| max {scale:.2f}", flush=True)
